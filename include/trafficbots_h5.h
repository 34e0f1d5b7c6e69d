/* C ABI of the packed-h5 scene reader: libtrafficbots_h5.so.
 *
 * Replaces, for the hot path's inputs, what the reference does per sample in Python workers with h5py
 * (`src/data_modules/data_h5_womd.py`): `DatasetBase.__init__` (:10-18, the "data_len" attribute), `DatasetVal.__getitem__`
 * (:38-55: the four episode attributes, one `hf[idx][key]` read per tensor, the all-ones dummy for agent tensors whose stored shape
 * differs from the configured one) and `DatasetTrain.__getitem__` (:27-35), followed by the DataLoader's collate (:229-241) and
 * the dtype half of the on-device pre-processing (`scene_centric.py:103-133`): the reader decodes every tensor of a BATCH of
 * episodes straight into one caller-provided (normally pinned) host buffer in the layout `tb_scene` / `tb_posterior_io` take
 * (include/trafficbots_hip.h): float32 values, uint8 masks, int32 class indices instead of bool one-hot rows.
 *
 * The file format is the one `src/pack_h5_womd.py:235,378-392` writes: root attribute "data_len"; one group per episode named by
 * its decimal index with attributes "scenario_id" (string), "scenario_center", "scenario_yaw", "with_map"; one chunked,
 * shuffle+gzip dataset per tensor, numpy bool stored as an 8-bit enum {FALSE, TRUE}.  The writer half (tb_h5_writer_*) produces
 * such files from synthetic scenes for the tests, tools/pack_synth_h5.py and bench-style runs without the Waymo data.
 *
 * Host-only: no HIP here.  Every function returns 0 on success, a negative code otherwise; tb_h5_last_error() explains.
 * One handle per calling thread; concurrent calls on different handles are safe when the HDF5 runtime is a thread-safe build (this
 * image's is: one library-wide lock) -- only the HDF5 metadata walk of tb_h5_read_batch is serialised by it, not the decoding.
 */
#ifndef TRAFFICBOTS_H5_H
#define TRAFFICBOTS_H5_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tb_h5_file tb_h5_file;
typedef struct tb_h5_writer tb_h5_writer;

/* how one stored tensor is decoded */
enum {
    TB_H5_F32 = 0,        /* any float/integer dataset -> float32 */
    TB_H5_MASK_U8 = 1,    /* bool (8-bit enum or integer) -> uint8 0/1 */
    TB_H5_ONEHOT_I32 = 2, /* bool one-hot [..., C] -> int32 [...]: index of the first set class, -1 when none is set */
    TB_H5_I64 = 3         /* integer dataset -> int64 */
};

#define TB_H5_ERR_ARG -1
#define TB_H5_ERR_IO -2    /* file / group / dataset / attribute missing or unreadable */
#define TB_H5_ERR_SHAPE -3 /* stored shape differs from the requested one and no dummy was allowed */

const char* tb_h5_last_error(void);

/* data_h5_womd.py:10-18.  Opens read-only (SWMR read when the file allows it, as the reference asks for). */
int32_t tb_h5_open(const char* path, tb_h5_file** out);
void tb_h5_close(tb_h5_file* f);
int64_t tb_h5_len(const tb_h5_file* f);

/* data_h5_womd.py:41-46.  scenario_id: NUL-terminated, truncated to id_cap-1 bytes; center: up to 3 values, n_center = how many the
 * file holds. */
int32_t tb_h5_episode_attrs(tb_h5_file* f, int64_t episode, char* scenario_id, int32_t id_cap, double center[3], int32_t* n_center,
                            double* yaw, int32_t* with_map);

/* the same for a batch in one call: scenario_ids [n_episode, id_cap] bytes, centers [n_episode, 3], the rest [n_episode] */
int32_t tb_h5_batch_attrs(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, char* scenario_ids, int32_t id_cap, double* centers,
                          int32_t* n_center, double* yaws, int32_t* with_maps);

/* Stored shape of "<episode>/<key>": rank (<= 8) and dims; elem_size in bytes. */
int32_t tb_h5_dataset_shape(tb_h5_file* f, int64_t episode, const char* key, int32_t* rank, int64_t dims[8], int32_t* elem_size);

/* data_h5_womd.py:47-52 for one key over a batch: out[e] = decode(hf[str(episodes[e])][key])[:n_lead], e = 0..n_episode-1,
 * contiguous (tb_h5_read_batch with one spec and one thread).  dims/rank = the configured shape of the stored tensor
 * (`tensor_size[key]`, rank >= 1); n_lead = number of leading entries of dim 0 to keep (0 = all; 11 turns a 91-step ground-truth tensor into its history part); for TB_H5_ONEHOT_I32 the last dim is
 * reduced away.  dummy_on_mismatch != 0: an episode whose stored shape differs yields all-ones (decoded: 1.0f / 1 / class 0 / 1),
 * the reference's behaviour for "agent" keys when n_agent is overridden (:50-52); otherwise TB_H5_ERR_SHAPE. */
int32_t tb_h5_read_key(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, const char* key, const int64_t* dims, int32_t rank,
                       int32_t n_lead, int32_t kind, int32_t dummy_on_mismatch, void* out);

/* One tensor of a batch read: the arguments of tb_h5_read_key (rank >= 1). */
typedef struct tb_h5_key_spec {
    const char* key;
    const int64_t* dims;
    int32_t rank;
    int32_t n_lead;
    int32_t kind;
    int32_t dummy_on_mismatch;
    void* out; /* [n_episode, decoded shape] */
} tb_h5_key_spec;

/* All tensors of a batch in one call: what `DataLoader(num_workers=...)` + collate do with worker PROCESSES (data_h5_womd.py:229-241),
 * done with threads inside one call.  The calling thread walks the HDF5 metadata (the library serialises its callers) and lists every
 * chunk's file extent; n_threads workers then pread, inflate (zlib), un-shuffle, scatter and decode the chunks without touching the
 * HDF5 library.  Tensors stored in any other form than the packer's (float32 / int64 / 1-byte bool, little endian, chunked or
 * contiguous, filters none / deflate / shuffle+deflate) are read through H5Dread by the calling thread instead.  n_threads <= 0:
 * everything through H5Dread. */
int32_t tb_h5_read_batch(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, const tb_h5_key_spec* specs, int32_t n_spec,
                         int32_t n_threads);

/* tb_h5_read_batch remembers, per handle, the stored form and chunk extents of every (episode, key) it has visited, so that the next
 * visit of that episode (the next validation epoch, training's random re-draws) makes no HDF5 call at all.  max_entries bounds that
 * index (default 2^20 tensors, about 280 bytes each); 0 disables and clears it.  The file must not change while it is open. */
int32_t tb_h5_set_index_cache(tb_h5_file* f, int64_t max_entries);
/* The same index on disk, so that the FIRST pass of a later run is metadata-free too.  save: writes the handle's index (merge_existing
 * != 0: after adding the entries an index file at `path` already holds, e.g. those of another reader's handle) atomically (temp file +
 * rename); load: adds the file's entries to the handle's index -- TB_H5_ERR_SHAPE and nothing loaded when the index was made for a data
 * file of another size or modification time. */
int32_t tb_h5_save_index(tb_h5_file* f, const char* path, int32_t merge_existing);
int32_t tb_h5_load_index(tb_h5_file* f, const char* path);

/* ---- writer (pack_h5_womd.py:235,378-392) ---- */
int32_t tb_h5_writer_open(const char* path, tb_h5_writer** out);
/* starts group str(episode) and writes its four attributes */
int32_t tb_h5_writer_episode(tb_h5_writer* w, int64_t episode, const char* scenario_id, const double* center, int32_t n_center, double yaw,
                             int32_t with_map);
/* storage of the datasets written from now on.  Default (the packer's): deflate_level 4, shuffle on, chunk_div 1 = one chunk per
 * tensor; chunk_div d > 1 splits every dim into d chunks (ceil(dim / d) rows each, the last one partial -- the shapes h5py's
 * automatic chunking produces for larger tensors); chunk_div 0 = contiguous, unfiltered. */
int32_t tb_h5_writer_options(tb_h5_writer* w, int32_t deflate_level, int32_t shuffle, int32_t chunk_div);
/* one tensor of the current episode; kind: TB_H5_F32 (float32 data), TB_H5_MASK_U8 (uint8 0/1 data, stored as the bool enum),
 * TB_H5_I64 (int64 data); chunked, shuffle + gzip level 4 */
int32_t tb_h5_writer_dataset(tb_h5_writer* w, const char* key, int32_t kind, const int64_t* dims, int32_t rank, const void* data);
/* writes "data_len" and closes */
int32_t tb_h5_writer_close(tb_h5_writer* w, int64_t data_len);

#ifdef __cplusplus
}
#endif
#endif
