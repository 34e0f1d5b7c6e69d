// Packed-h5 scene reader / writer behind include/trafficbots_h5.h.  Host code over the HDF5 C library; decodes
// the tensors of a batch of episodes into the layout the HIP entry points take, so that the only work left after the read is
// the host-to-device copy.  Format and reference behaviour: `src/data_modules/data_h5_womd.py:10-55`, `src/pack_h5_womd.py:378-392`.
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/trafficbots_h5.h"

namespace {

thread_local std::string g_err;

// The HDF5 library takes its own lock per API call; two reader threads walking metadata at once hand that lock back and forth
// thousands of times per batch and both get slower (measured: 1000 -> 650 episodes/s).  Whole metadata walks are therefore
// serialised here, coarsely; the decode phase runs outside.
std::mutex g_hdf5_walk;

int32_t fail(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

// RAII for HDF5 ids
struct Hid {
    hid_t id;
    herr_t (*closer)(hid_t);
    Hid(hid_t i, herr_t (*c)(hid_t)) : id(i), closer(c) {}
    ~Hid() {
        if (id >= 0) closer(id);
    }
    Hid(const Hid&) = delete;
    Hid& operator=(const Hid&) = delete;
    operator hid_t() const { return id; }
    bool ok() const { return id >= 0; }
};

// errors are reported through return codes + tb_h5_last_error; the automatic stack printing is a per-thread setting in a
// thread-safe HDF5 build, so every entry point switches it off for its calling thread
void quiet() {
    static thread_local bool done = false;
    if (!done) {
        H5open();
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        done = true;
    }
}

int64_t count_of(const int64_t* dims, int32_t rank) {
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    return n;
}

size_t out_elem_size(int32_t kind) {
    switch (kind) {
        case TB_H5_F32: return 4;
        case TB_H5_MASK_U8: return 1;
        case TB_H5_ONEHOT_I32: return 4;
        default: return 8;
    }
}

}  // namespace

struct tb_h5_file {
    hid_t file = -1;
    int64_t len = 0;
    int fd = -1;       // second, plain descriptor of the same file: chunk bytes are pread() outside the HDF5 library
    haddr_t base = 0;  // user block size: chunk addresses are relative to it
    struct tb_h5_index* index = nullptr;  // stored forms + chunk extents of the tensors visited so far
};

struct tb_h5_index* tb_h5_new_index();
void tb_h5_free_index(struct tb_h5_index*);

struct tb_h5_writer {
    hid_t file = -1;
    hid_t group = -1;
    hid_t bool_type = -1;
    hid_t lcpl = -1;
    int deflate = 4, shuffle = 1, chunk_div = 1;
};

extern "C" {

const char* tb_h5_last_error(void) { return g_err.c_str(); }

int32_t tb_h5_open(const char* path, tb_h5_file** out) {
    if (!path || !out) return fail(TB_H5_ERR_ARG, "tb_h5_open: null argument");
    quiet();
    Hid fapl(H5Pcreate(H5P_FILE_ACCESS), H5Pclose);
    H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
    hid_t file = H5Fopen(path, H5F_ACC_RDONLY | H5F_ACC_SWMR_READ, fapl);
    if (file < 0) file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);  // files not written for SWMR: a plain read-only open
    if (file < 0) return fail(TB_H5_ERR_IO, std::string("tb_h5_open: cannot open ") + path);
    int64_t len = -1;
    {
        Hid a(H5Aexists(file, "data_len") > 0 ? H5Aopen(file, "data_len", H5P_DEFAULT) : -1, H5Aclose);
        if (a.ok()) {
            if (H5Aread(a, H5T_NATIVE_INT64, &len) < 0 || len < 0) {
                H5Fclose(file);
                return fail(TB_H5_ERR_IO, std::string("tb_h5_open: unreadable 'data_len' attribute in ") + path);
            }
        } else {
            // a file without the packer's root attribute (e.g. re-assembled with the HDF5 command-line tools, which cannot write
            // attributes): the episodes are the root groups "0" .. "n-1" -- count the consecutive ones
            len = 0;
            while (H5Lexists(file, std::to_string(len).c_str(), H5P_DEFAULT) > 0) ++len;
            if (len == 0) {
                H5Fclose(file);
                return fail(TB_H5_ERR_IO, std::string("tb_h5_open: no 'data_len' attribute and no episode group \"0\" in ") + path);
            }
        }
    }
    tb_h5_file* f = new tb_h5_file();
    f->file = file;
    f->len = len;
    f->fd = open(path, O_RDONLY);
    f->index = tb_h5_new_index();
    {
        Hid fcpl(H5Fget_create_plist(file), H5Pclose);
        hsize_t ub = 0;
        if (fcpl.ok() && H5Pget_userblock(fcpl, &ub) >= 0) f->base = (haddr_t)ub;
    }
    *out = f;
    return 0;
}

void tb_h5_close(tb_h5_file* f) {
    if (!f) return;
    if (f->file >= 0) H5Fclose(f->file);
    if (f->fd >= 0) close(f->fd);
    tb_h5_free_index(f->index);
    delete f;
}

int64_t tb_h5_len(const tb_h5_file* f) { return f ? f->len : -1; }

int32_t tb_h5_episode_attrs(tb_h5_file* f, int64_t episode, char* scenario_id, int32_t id_cap, double center[3], int32_t* n_center,
                            double* yaw, int32_t* with_map) {
    if (!f) return fail(TB_H5_ERR_ARG, "tb_h5_episode_attrs: null handle");
    quiet();
    const std::string name = std::to_string(episode);
    Hid g(H5Gopen2(f->file, name.c_str(), H5P_DEFAULT), H5Gclose);
    if (!g.ok()) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: no episode group " + name);
    if (scenario_id && id_cap > 0) {
        scenario_id[0] = 0;
        Hid a(H5Aopen(g, "scenario_id", H5P_DEFAULT), H5Aclose);
        if (!a.ok()) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: no scenario_id in episode " + name);
        Hid ty(H5Aget_type(a), H5Tclose);
        if (H5Tget_class(ty) != H5T_STRING) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: scenario_id is not a string");
        if (H5Tis_variable_str(ty) > 0) {
            char* s = nullptr;
            Hid mt(H5Tcopy(H5T_C_S1), H5Tclose);
            H5Tset_size(mt, H5T_VARIABLE);
            H5Tset_cset(mt, H5Tget_cset(ty));
            if (H5Aread(a, mt, &s) < 0 || !s) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: cannot read scenario_id");
            snprintf(scenario_id, (size_t)id_cap, "%s", s);
            H5free_memory(s);
        } else {
            const size_t n = H5Tget_size(ty);
            std::vector<char> buf(n + 1, 0);
            if (H5Aread(a, ty, buf.data()) < 0) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: cannot read scenario_id");
            snprintf(scenario_id, (size_t)id_cap, "%s", buf.data());
        }
    }
    if (center && n_center) {
        Hid a(H5Aopen(g, "scenario_center", H5P_DEFAULT), H5Aclose);
        if (!a.ok()) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: no scenario_center in episode " + name);
        Hid sp(H5Aget_space(a), H5Sclose);
        const hssize_t n = H5Sget_simple_extent_npoints(sp);
        if (n < 1 || n > 3) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: scenario_center holds neither 1, 2 nor 3 values");
        double tmp[3] = {0, 0, 0};
        if (H5Aread(a, H5T_NATIVE_DOUBLE, tmp) < 0) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: cannot read scenario_center");
        for (int i = 0; i < 3; ++i) center[i] = tmp[i];
        *n_center = (int32_t)n;
    }
    if (yaw) {
        Hid a(H5Aopen(g, "scenario_yaw", H5P_DEFAULT), H5Aclose);
        if (!a.ok() || H5Aread(a, H5T_NATIVE_DOUBLE, yaw) < 0)
            return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: no readable scenario_yaw in episode " + name);
    }
    if (with_map) {
        Hid a(H5Aopen(g, "with_map", H5P_DEFAULT), H5Aclose);
        if (!a.ok()) return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: no with_map in episode " + name);
        Hid ty(H5Aget_type(a), H5Tclose);
        Hid nt(H5Tget_native_type(ty, H5T_DIR_ASCEND), H5Tclose);
        uint8_t raw[16] = {0};
        if (H5Tget_size(nt) > sizeof(raw) || H5Aread(a, nt, raw) < 0)
            return fail(TB_H5_ERR_IO, "tb_h5_episode_attrs: cannot read with_map");
        int any = 0;
        for (size_t i = 0; i < H5Tget_size(nt); ++i) any |= raw[i];
        *with_map = any != 0;
    }
    return 0;
}

int32_t tb_h5_batch_attrs(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, char* scenario_ids, int32_t id_cap, double* centers,
                          int32_t* n_center, double* yaws, int32_t* with_maps) {
    if (!f || !episodes || !scenario_ids || id_cap < 1 || !centers || !n_center || !yaws || !with_maps)
        return fail(TB_H5_ERR_ARG, "tb_h5_batch_attrs: null argument");
    std::lock_guard<std::mutex> walk(g_hdf5_walk);
    for (int32_t e = 0; e < n_episode; ++e) {
        const int32_t rc = tb_h5_episode_attrs(f, episodes[e], scenario_ids + (size_t)e * (size_t)id_cap, id_cap, centers + 3 * e, n_center + e,
                                               yaws + e, with_maps + e);
        if (rc) return rc;
    }
    return 0;
}

int32_t tb_h5_dataset_shape(tb_h5_file* f, int64_t episode, const char* key, int32_t* rank, int64_t dims[8], int32_t* elem_size) {
    if (!f || !key || !rank || !dims) return fail(TB_H5_ERR_ARG, "tb_h5_dataset_shape: null argument");
    quiet();
    const std::string path = std::to_string(episode) + "/" + key;
    Hid d(H5Dopen2(f->file, path.c_str(), H5P_DEFAULT), H5Dclose);
    if (!d.ok()) return fail(TB_H5_ERR_IO, "tb_h5_dataset_shape: no dataset " + path);
    Hid sp(H5Dget_space(d), H5Sclose);
    const int r = H5Sget_simple_extent_ndims(sp);
    if (r < 0 || r > 8) return fail(TB_H5_ERR_IO, "tb_h5_dataset_shape: rank of " + path + " is not in 0..8");
    hsize_t hd[8] = {0};
    H5Sget_simple_extent_dims(sp, hd, nullptr);
    *rank = r;
    for (int i = 0; i < r; ++i) dims[i] = (int64_t)hd[i];
    if (elem_size) {
        Hid ty(H5Dget_type(d), H5Tclose);
        *elem_size = (int32_t)H5Tget_size(ty);
    }
    return 0;
}

}  // extern "C"

// ---- batch read ---------------------------------------------------------------------------------------------------------------
// Phase 1 (this thread, inside the HDF5 library, which serialises its callers): per (episode, key) open the dataset, check shape and
// storage type, and list its chunks as (file address, stored bytes, chunk coordinate).  Phase 2 (n_threads workers, no HDF5 calls):
// pread each chunk from the file, inflate, un-shuffle, scatter the wanted rows into place; the worker that lands the last chunk
// of a tensor decodes it (mask normalisation / one-hot -> class index).  Datasets whose storage is not one of the expected forms
// (float32 / int64 / 1-byte bool, little endian, chunked or contiguous, filters {} / {deflate} / {shuffle, deflate}) take the
// library's own H5Dread in phase 1 instead.
namespace {

struct DsTask {
    int rank = 0;
    hsize_t dd[8] = {0}, cd[8] = {0}, lim[8] = {0};  // dataset dims, chunk dims, rows wanted per dim (lim[0] = n_lead)
    size_t esz = 0;
    int kind = 0;
    bool shuffle = false, deflate = false;
    int shuffle_bit = -1, deflate_bit = -1;  // positions in the filter pipeline (bits of a chunk's filter mask)
    uint8_t* dst = nullptr;                  // decoded output of this (episode, key)
    uint8_t* stage = nullptr;                // stored-type elements are assembled here (== dst unless a class reduction follows)
    int64_t n_class = 1, n_out = 0, n_read = 0;
    std::atomic<int> remaining{0};
    std::string path;
};

struct ChunkTask {
    DsTask* ds;
    haddr_t addr;
    hsize_t nbytes;
    unsigned filter_mask;
    hsize_t coord[8];
};

void decode_in_place(DsTask& t) {
    if (t.kind == TB_H5_MASK_U8) {
        for (int64_t i = 0; i < t.n_out; ++i) t.dst[i] = t.dst[i] != 0;
    } else if (t.kind == TB_H5_ONEHOT_I32) {
        int32_t* o = (int32_t*)t.dst;
        for (int64_t i = 0; i < t.n_out; ++i) {
            const uint8_t* row = t.stage + i * t.n_class;
            int32_t c = -1;
            for (int64_t j = 0; j < t.n_class; ++j)
                if (row[j]) {
                    c = (int32_t)j;
                    break;
                }
            o[i] = c;
        }
    }
}

// copy the part of one chunk that lies inside [0, lim) into the row-major [lim[0], dd[1], ..., dd[rank-1]] staging tensor
void scatter_chunk(const DsTask& t, const uint8_t* chunk, const hsize_t* coord) {
    const int r = t.rank;
    hsize_t ext[8];
    for (int i = 0; i < r; ++i) {
        const hsize_t hi = coord[i] + t.cd[i] < t.lim[i] ? coord[i] + t.cd[i] : t.lim[i];
        if (hi <= coord[i]) return;
        ext[i] = hi - coord[i];
    }
    const size_t run = (size_t)ext[r - 1] * t.esz;
    hsize_t idx[8] = {0};
    for (;;) {
        size_t src = 0, dst = 0;
        for (int i = 0; i < r - 1; ++i) {
            src = (src + idx[i]) * t.cd[i + 1];
            dst = (dst + coord[i] + idx[i]) * t.dd[i + 1];
        }
        dst += coord[r - 1];
        memcpy(t.stage + dst * t.esz, chunk + src * t.esz, run);
        int d = r - 2;
        while (d >= 0 && ++idx[d] == ext[d]) idx[d--] = 0;
        if (d < 0) break;
    }
}

struct WorkerScratch {
    std::vector<uint8_t> raw, plain, unshuffled;
};

bool run_chunk(int fd, haddr_t base, const ChunkTask& c, WorkerScratch& ws, std::string& err) {
    DsTask& t = *c.ds;
    size_t chunk_bytes = t.esz;
    for (int i = 0; i < t.rank; ++i) chunk_bytes *= (size_t)t.cd[i];
    ws.raw.resize((size_t)c.nbytes);
    size_t got = 0;
    while (got < c.nbytes) {
        const ssize_t n = pread(fd, ws.raw.data() + got, (size_t)c.nbytes - got, (off_t)(base + c.addr + got));
        if (n <= 0) {
            err = "short read of a chunk of " + t.path;
            return false;
        }
        got += (size_t)n;
    }
    const uint8_t* cur = ws.raw.data();
    if (t.deflate && !(c.filter_mask & (1u << t.deflate_bit))) {
        ws.plain.resize(chunk_bytes);
        uLongf n = (uLongf)chunk_bytes;
        if (uncompress(ws.plain.data(), &n, cur, (uLong)c.nbytes) != Z_OK || n != chunk_bytes) {
            err = "inflate of a chunk of " + t.path + " failed";
            return false;
        }
        cur = ws.plain.data();
    } else if (c.nbytes != chunk_bytes) {
        err = "unexpected stored size of a chunk of " + t.path;
        return false;
    }
    if (t.shuffle && t.esz > 1 && !(c.filter_mask & (1u << t.shuffle_bit))) {
        ws.unshuffled.resize(chunk_bytes);
        const size_t n = chunk_bytes / t.esz;
        uint8_t* o = ws.unshuffled.data();
        for (size_t j = 0; j < t.esz; ++j) {
            const uint8_t* plane = cur + j * n;
            for (size_t i = 0; i < n; ++i) o[i * t.esz + j] = plane[i];
        }
        cur = o;
    }
    scatter_chunk(t, cur, c.coord);
    if (t.remaining.fetch_sub(1) == 1) decode_in_place(t);
    return true;
}

// the library's own read + conversion of one tensor (any storage)
int32_t read_generic(tb_h5_file* f, hid_t d, hid_t sp, DsTask& t) {
    if (t.n_read == 0) return 0;
    hsize_t zero[8] = {0};
    Hid msp(H5Screate_simple(t.rank, t.lim, nullptr), H5Sclose);
    if (H5Sselect_hyperslab(sp, H5S_SELECT_SET, zero, nullptr, t.lim, nullptr) < 0)
        return fail(TB_H5_ERR_IO, "tb_h5_read: cannot select the leading rows of " + t.path);
    herr_t rc;
    if (t.kind == TB_H5_F32) rc = H5Dread(d, H5T_NATIVE_FLOAT, msp, sp, H5P_DEFAULT, t.dst);
    else if (t.kind == TB_H5_I64) rc = H5Dread(d, H5T_NATIVE_INT64, msp, sp, H5P_DEFAULT, t.dst);
    else {
        Hid ty(H5Dget_type(d), H5Tclose);
        const H5T_class_t cls = H5Tget_class(ty);
        if (cls == H5T_ENUM && H5Tget_size(ty) == 1) rc = H5Dread(d, ty, msp, sp, H5P_DEFAULT, t.stage);
        else if (cls == H5T_INTEGER) rc = H5Dread(d, H5T_NATIVE_UINT8, msp, sp, H5P_DEFAULT, t.stage);
        else return fail(TB_H5_ERR_IO, "tb_h5_read: " + t.path + " is neither a bool enum nor an integer dataset");
    }
    if (rc < 0) return fail(TB_H5_ERR_IO, "tb_h5_read: read of " + t.path + " failed");
    decode_in_place(t);
    (void)f;
    return 0;
}

}  // namespace

namespace {

// what phase 1 learns about one stored tensor; kept per (episode, key) in the handle's index so that a later visit of the same
// episode (next epoch, training's random re-draws) costs no HDF5 call at all
struct ChunkRec {
    haddr_t addr;
    hsize_t nbytes;
    unsigned filter_mask;
    hsize_t coord[8];
};
struct StoredForm {
    int rank = -1;
    hsize_t dd[8] = {0}, cd[8] = {0};
    size_t esz = 0;
    H5T_class_t cls = H5T_NO_CLASS;
    bool little = false, direct = false;  // direct: chunk extents known, storage in one of the expected forms
    bool shuffle = false, deflate = false;
    int shuffle_bit = -1, deflate_bit = -1;
    hsize_t n_expected = 0;
    std::vector<ChunkRec> chunks;
};

bool probe(hid_t d, hid_t sp, StoredForm& sf) {
    sf.rank = H5Sget_simple_extent_ndims(sp);
    if (sf.rank < 0 || sf.rank > 8) return false;
    H5Sget_simple_extent_dims(sp, sf.dd, nullptr);
    Hid ty(H5Dget_type(d), H5Tclose);
    sf.cls = H5Tget_class(ty);
    sf.esz = H5Tget_size(ty);
    sf.little = sf.esz == 1 || H5Tget_order(ty) == H5T_ORDER_LE;
    Hid dcpl(H5Dget_create_plist(d), H5Pclose);
    const H5D_layout_t layout = H5Pget_layout(dcpl);
    const int nf = H5Pget_nfilters(dcpl);
    bool ok = sf.rank >= 1;
    for (int i = 0; ok && i < nf; ++i) {
        unsigned flags = 0, cfg = 0;
        size_t nel = 0;
        const H5Z_filter_t id = H5Pget_filter2(dcpl, (unsigned)i, &flags, &nel, nullptr, 0, nullptr, &cfg);
        if (id == H5Z_FILTER_SHUFFLE && !sf.shuffle && !sf.deflate) sf.shuffle = true, sf.shuffle_bit = i;
        else if (id == H5Z_FILTER_DEFLATE && !sf.deflate) sf.deflate = true, sf.deflate_bit = i;
        else ok = false;
    }
    if (ok && layout == H5D_CHUNKED) {
        H5Pget_chunk(dcpl, sf.rank, sf.cd);
        hsize_t n_chunk = 0;
        sf.n_expected = 1;
        for (int i = 0; i < sf.rank; ++i) sf.n_expected *= sf.cd[i] ? (sf.dd[i] + sf.cd[i] - 1) / sf.cd[i] : 0;
        ok = H5Dget_num_chunks(d, sp, &n_chunk) >= 0;
        for (hsize_t c = 0; ok && c < n_chunk; ++c) {
            ChunkRec r;
            ok = H5Dget_chunk_info(d, sp, c, r.coord, &r.filter_mask, &r.addr, &r.nbytes) >= 0 && r.addr != HADDR_UNDEF;
            if (ok) sf.chunks.push_back(r);
        }
    } else if (ok && layout == H5D_CONTIGUOUS && nf == 0) {
        ChunkRec r;
        r.addr = H5Dget_offset(d);
        r.filter_mask = 0;
        r.nbytes = sf.esz;
        for (int i = 0; i < sf.rank; ++i) sf.cd[i] = sf.dd[i], r.coord[i] = 0, r.nbytes *= sf.dd[i];
        sf.n_expected = 1;
        ok = r.addr != HADDR_UNDEF && r.nbytes > 0;
        if (ok) sf.chunks.push_back(r);
    } else {
        ok = false;
    }
    sf.direct = ok;
    if (!ok) sf.chunks.clear();
    return true;
}

}  // namespace

struct tb_h5_index {
    std::unordered_map<std::string, StoredForm> map;
    size_t max_entries = (size_t)1 << 20;
};

tb_h5_index* tb_h5_new_index() { return new tb_h5_index(); }
void tb_h5_free_index(tb_h5_index* p) { delete p; }

extern "C" int32_t tb_h5_set_index_cache(tb_h5_file* f, int64_t max_entries) {
    if (!f || max_entries < 0) return fail(TB_H5_ERR_ARG, "tb_h5_set_index_cache: bad argument");
    f->index->max_entries = (size_t)max_entries;
    if (f->index->map.size() > (size_t)max_entries) f->index->map.clear();
    return 0;
}

// ---- chunk index on disk: [magic "TBH5IDX1"][file size][file mtime ns][n entries] then per entry the path and the StoredForm
namespace {

const char IDX_MAGIC[8] = {'T', 'B', 'H', '5', 'I', 'D', 'X', '1'};

bool file_stamp(int fd, int64_t stamp[2]) {
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) return false;
    stamp[0] = (int64_t)st.st_size;
    stamp[1] = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}

template <class T>
bool put(FILE* fp, const T& v) {
    return fwrite(&v, sizeof(T), 1, fp) == 1;
}
template <class T>
bool get(FILE* fp, T& v) {
    return fread(&v, sizeof(T), 1, fp) == 1;
}

}  // namespace

extern "C" int32_t tb_h5_save_index(tb_h5_file* f, const char* path, int32_t merge_existing) {
    if (!f || !path) return fail(TB_H5_ERR_ARG, "tb_h5_save_index: null argument");
    if (merge_existing) (void)tb_h5_load_index(f, path);  // entries of other handles / earlier runs; a mismatching file is ignored
    int64_t stamp[2];
    if (!file_stamp(f->fd, stamp)) return fail(TB_H5_ERR_IO, "tb_h5_save_index: cannot stat the data file");
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid());
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return fail(TB_H5_ERR_IO, "tb_h5_save_index: cannot create " + tmp);
    bool ok = fwrite(IDX_MAGIC, 8, 1, fp) == 1 && put(fp, stamp[0]) && put(fp, stamp[1]) && put(fp, (uint64_t)f->index->map.size());
    for (auto it = f->index->map.begin(); ok && it != f->index->map.end(); ++it) {
        const StoredForm& sf = it->second;
        const uint32_t len = (uint32_t)it->first.size(), nch = (uint32_t)sf.chunks.size();
        const int32_t head[8] = {sf.rank, (int32_t)sf.esz, (int32_t)sf.cls, sf.little, sf.direct, sf.shuffle | (sf.deflate << 1), sf.shuffle_bit,
                                 sf.deflate_bit};
        ok = put(fp, len) && fwrite(it->first.data(), 1, len, fp) == len && fwrite(head, sizeof(head), 1, fp) == 1 &&
             fwrite(sf.dd, sizeof(sf.dd), 1, fp) == 1 && fwrite(sf.cd, sizeof(sf.cd), 1, fp) == 1 && put(fp, sf.n_expected) && put(fp, nch) &&
             (nch == 0 || fwrite(sf.chunks.data(), sizeof(ChunkRec), nch, fp) == nch);
    }
    ok = (fclose(fp) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        remove(tmp.c_str());
        return fail(TB_H5_ERR_IO, std::string("tb_h5_save_index: cannot write ") + path);
    }
    return 0;
}

extern "C" int32_t tb_h5_load_index(tb_h5_file* f, const char* path) {
    if (!f || !path) return fail(TB_H5_ERR_ARG, "tb_h5_load_index: null argument");
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(TB_H5_ERR_IO, std::string("tb_h5_load_index: cannot open ") + path);
    struct Closer {
        FILE* fp;
        ~Closer() { fclose(fp); }
    } closer{fp};
    char magic[8];
    int64_t stamp[2], have[2];
    uint64_t n = 0;
    if (fread(magic, 8, 1, fp) != 1 || memcmp(magic, IDX_MAGIC, 8) != 0 || !get(fp, stamp[0]) || !get(fp, stamp[1]) || !get(fp, n))
        return fail(TB_H5_ERR_IO, std::string("tb_h5_load_index: not an index file: ") + path);
    if (!file_stamp(f->fd, have) || have[0] != stamp[0] || have[1] != stamp[1])
        return fail(TB_H5_ERR_SHAPE, std::string("tb_h5_load_index: index was made for another version of the data file: ") + path);
    std::unordered_map<std::string, StoredForm> fresh;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t len = 0, nch = 0;
        int32_t head[8];
        StoredForm sf;
        std::string key;
        bool ok = get(fp, len) && len < 4096;
        if (ok) {
            key.resize(len);
            ok = fread(&key[0], 1, len, fp) == len && fread(head, sizeof(head), 1, fp) == 1 && fread(sf.dd, sizeof(sf.dd), 1, fp) == 1 &&
                 fread(sf.cd, sizeof(sf.cd), 1, fp) == 1 && get(fp, sf.n_expected) && get(fp, nch) && nch < (1u << 24);
        }
        if (ok) {
            sf.rank = head[0], sf.esz = (size_t)head[1], sf.cls = (H5T_class_t)head[2], sf.little = head[3] != 0, sf.direct = head[4] != 0;
            sf.shuffle = head[5] & 1, sf.deflate = (head[5] >> 1) & 1, sf.shuffle_bit = head[6], sf.deflate_bit = head[7];
            sf.chunks.resize(nch);
            ok = nch == 0 || fread(sf.chunks.data(), sizeof(ChunkRec), nch, fp) == nch;
        }
        if (!ok) return fail(TB_H5_ERR_IO, std::string("tb_h5_load_index: truncated index file: ") + path);
        fresh.emplace(std::move(key), std::move(sf));
    }
    for (auto& kv : fresh)
        if (f->index->map.size() < f->index->max_entries) f->index->map.emplace(kv.first, std::move(kv.second));
    return 0;
}

extern "C" int32_t tb_h5_read_batch(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, const tb_h5_key_spec* specs, int32_t n_spec,
                                    int32_t n_threads) {
    if (!f || !episodes || !specs || n_episode < 0 || n_spec < 0) return fail(TB_H5_ERR_ARG, "tb_h5_read_batch: bad argument");
    quiet();
    std::unique_lock<std::mutex> walk(g_hdf5_walk);
    const auto t_begin = std::chrono::steady_clock::now();
    std::deque<DsTask> tasks;  // stable addresses
    std::vector<ChunkTask> chunks;
    std::vector<std::vector<uint8_t>> stages;
    struct Groups {  // the episode groups, opened on first need, once per batch
        std::vector<hid_t> ids;
        ~Groups() {
            for (hid_t g : ids)
                if (g >= 0) H5Gclose(g);
        }
    } groups;
    groups.ids.assign((size_t)n_episode, -1);
    size_t n_probed = 0;
    for (int32_t k = 0; k < n_spec; ++k) {
        const tb_h5_key_spec& sp_ = specs[k];
        const int rank = sp_.rank;
        if (!sp_.key || !sp_.out || rank < 1 || rank > 8 || !sp_.dims || sp_.kind < TB_H5_F32 || sp_.kind > TB_H5_I64 || sp_.n_lead < 0 ||
            sp_.n_lead > sp_.dims[0])
            return fail(TB_H5_ERR_ARG, std::string("tb_h5_read_batch: bad spec for key ") + (sp_.key ? sp_.key : "(null)"));
        int64_t n_read = 1;
        for (int i = 0; i < rank; ++i) n_read *= i == 0 && sp_.n_lead > 0 ? sp_.n_lead : sp_.dims[i];
        const int64_t n_class = sp_.kind == TB_H5_ONEHOT_I32 ? sp_.dims[rank - 1] : 1;
        const int64_t n_out = n_class ? n_read / n_class : 0;
        const size_t osz = out_elem_size(sp_.kind);
        for (int32_t e = 0; e < n_episode; ++e) {
            tasks.emplace_back();
            DsTask& t = tasks.back();
            t.rank = rank, t.kind = sp_.kind, t.n_class = n_class, t.n_out = n_out, t.n_read = n_read;
            t.dst = (uint8_t*)sp_.out + (size_t)e * (size_t)n_out * osz;
            t.path = std::to_string(episodes[e]) + "/" + sp_.key;
            for (int i = 0; i < rank; ++i) t.lim[i] = (hsize_t)(i == 0 && sp_.n_lead > 0 ? sp_.n_lead : sp_.dims[i]);
            // stored form: from the index, or from the file
            StoredForm local;
            const StoredForm* sf = nullptr;
            auto hit = f->index->map.find(t.path);
            if (hit != f->index->map.end()) sf = &hit->second;
            hid_t d_id = -1, sp_id = -1;
            auto open_ds = [&]() -> bool {
                if (groups.ids[e] < 0) groups.ids[e] = H5Gopen2(f->file, std::to_string(episodes[e]).c_str(), H5P_DEFAULT);
                if (groups.ids[e] < 0) return false;
                d_id = H5Dopen2(groups.ids[e], sp_.key, H5P_DEFAULT);
                if (d_id < 0) return false;
                sp_id = H5Dget_space(d_id);
                return sp_id >= 0;
            };
            Hid d_guard(-1, H5Dclose), sp_guard(-1, H5Sclose);
            if (!sf) {
                const bool opened = open_ds();
                d_guard.id = d_id, sp_guard.id = sp_id;
                if (!opened) return fail(TB_H5_ERR_IO, "tb_h5_read: no dataset " + t.path);
                if (!probe(d_id, sp_id, local)) return fail(TB_H5_ERR_IO, "tb_h5_read: rank of " + t.path + " is not in 0..8");
                ++n_probed;
                if (f->index->map.size() < f->index->max_entries) sf = &(f->index->map[t.path] = std::move(local));
                else sf = &local;
            }
            bool same = sf->rank == rank;
            for (int i = 0; same && i < rank; ++i) same = (int64_t)sf->dd[i] == sp_.dims[i];
            if (!same) {
                if (!sp_.dummy_on_mismatch)
                    return fail(TB_H5_ERR_SHAPE, "tb_h5_read: stored shape of " + t.path + " differs from the configured one");
                // np.ones(size, dtype) decoded: 1.0f, mask 1, every class set -> first class, integer 1
                if (t.kind == TB_H5_F32) for (int64_t i = 0; i < n_out; ++i) ((float*)t.dst)[i] = 1.f;
                else if (t.kind == TB_H5_MASK_U8) memset(t.dst, 1, (size_t)n_out);
                else if (t.kind == TB_H5_ONEHOT_I32) memset(t.dst, 0, (size_t)n_out * 4);
                else for (int64_t i = 0; i < n_out; ++i) ((int64_t*)t.dst)[i] = 1;
                continue;
            }
            if (n_read == 0) continue;
            if (t.kind == TB_H5_ONEHOT_I32) {
                stages.emplace_back((size_t)n_read);
                t.stage = stages.back().data();
            } else {
                t.stage = t.dst;
            }
            for (int i = 0; i < rank; ++i) t.dd[i] = sf->dd[i], t.cd[i] = sf->cd[i];
            t.esz = sf->esz;
            bool fast = sf->direct && f->fd >= 0 && n_threads > 0 && sf->little;
            if (t.kind == TB_H5_F32) fast = fast && sf->cls == H5T_FLOAT && sf->esz == 4;
            else if (t.kind == TB_H5_I64) fast = fast && sf->cls == H5T_INTEGER && sf->esz == 8;
            else fast = fast && (sf->cls == H5T_ENUM || sf->cls == H5T_INTEGER) && sf->esz == 1;
            if (!fast) {  // the library's own read + conversion
                if (d_id < 0) {
                    const bool opened = open_ds();
                    d_guard.id = d_id, sp_guard.id = sp_id;
                    if (!opened) return fail(TB_H5_ERR_IO, "tb_h5_read: no dataset " + t.path);
                }
                const int32_t rc = read_generic(f, d_id, sp_id, t);
                if (rc) return rc;
                continue;
            }
            t.shuffle = sf->shuffle, t.deflate = sf->deflate, t.shuffle_bit = sf->shuffle_bit, t.deflate_bit = sf->deflate_bit;
            if (sf->chunks.size() != sf->n_expected) memset(t.stage, 0, (size_t)n_read * t.esz);  // unwritten chunks read as the fill value
            int n_mine = 0;
            for (const ChunkRec& r : sf->chunks) {
                if (r.coord[0] >= t.lim[0]) continue;  // chunks wholly past the wanted leading rows are never touched
                ChunkTask ct;
                ct.ds = &t, ct.addr = r.addr, ct.nbytes = r.nbytes, ct.filter_mask = r.filter_mask;
                memcpy(ct.coord, r.coord, sizeof(ct.coord));
                chunks.push_back(ct);
                ++n_mine;
            }
            t.remaining.store(n_mine);
            if (n_mine == 0) decode_in_place(t);
        }
    }
    const size_t n_probed_total = n_probed;
    walk.unlock();
    if (chunks.empty()) return 0;
    const auto t_meta = std::chrono::steady_clock::now();
    // phase 2
    const int nt = std::max(1, std::min<int>(n_threads, (int)chunks.size()));
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::mutex mu;
    std::string err;
    auto work = [&]() {
        WorkerScratch ws;
        std::string e;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size() || bad.load()) return;
            if (!run_chunk(f->fd, f->base, chunks[i], ws, e)) {
                std::lock_guard<std::mutex> g(mu);
                if (!bad.exchange(true)) err = e;
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (getenv("TB_H5_DEBUG")) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "tb_h5_read_batch: %zu tensors (%zu probed in the file), %zu chunks, metadata %.2f ms, decode %.2f ms on %d threads\n", tasks.size(),
                n_probed_total, chunks.size(),
                std::chrono::duration<double, std::milli>(t_meta - t_begin).count(), std::chrono::duration<double, std::milli>(t_end - t_meta).count(), nt);
    }
    if (bad.load()) return fail(TB_H5_ERR_IO, "tb_h5_read: " + err);
    return 0;
}

extern "C" int32_t tb_h5_read_key(tb_h5_file* f, const int64_t* episodes, int32_t n_episode, const char* key, const int64_t* dims, int32_t rank,
                                  int32_t n_lead, int32_t kind, int32_t dummy_on_mismatch, void* out) {
    tb_h5_key_spec s;
    s.key = key, s.dims = dims, s.rank = rank, s.n_lead = n_lead, s.kind = kind, s.dummy_on_mismatch = dummy_on_mismatch, s.out = out;
    return tb_h5_read_batch(f, episodes, n_episode, &s, 1, 1);
}

extern "C" {

// ------------------------------------------------------------------------------------------------ writer

int32_t tb_h5_writer_open(const char* path, tb_h5_writer** out) {
    if (!path || !out) return fail(TB_H5_ERR_ARG, "tb_h5_writer_open: null argument");
    quiet();
    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (file < 0) return fail(TB_H5_ERR_IO, std::string("tb_h5_writer_open: cannot create ") + path);
    tb_h5_writer* w = new tb_h5_writer();
    w->file = file;
    w->bool_type = H5Tenum_create(H5T_NATIVE_INT8);
    int8_t v = 0;
    H5Tenum_insert(w->bool_type, "FALSE", &v);
    v = 1;
    H5Tenum_insert(w->bool_type, "TRUE", &v);
    w->lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(w->lcpl, 1);
    *out = w;
    return 0;
}

static int32_t write_attr(hid_t loc, const char* name, hid_t ftype, hid_t mtype, int n, const void* data) {
    hsize_t d = (hsize_t)n;
    Hid sp(n < 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(1, &d, nullptr), H5Sclose);
    Hid a(H5Acreate2(loc, name, ftype, sp, H5P_DEFAULT, H5P_DEFAULT), H5Aclose);
    if (!a.ok() || H5Awrite(a, mtype, data) < 0) return fail(TB_H5_ERR_IO, std::string("tb_h5_writer: cannot write attribute ") + name);
    return 0;
}

int32_t tb_h5_writer_episode(tb_h5_writer* w, int64_t episode, const char* scenario_id, const double* center, int32_t n_center, double yaw,
                             int32_t with_map) {
    if (!w || !scenario_id || (n_center > 0 && !center)) return fail(TB_H5_ERR_ARG, "tb_h5_writer_episode: null argument");
    if (w->group >= 0) H5Gclose(w->group);
    const std::string name = std::to_string(episode);
    w->group = H5Gcreate2(w->file, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (w->group < 0) return fail(TB_H5_ERR_IO, "tb_h5_writer_episode: cannot create group " + name);
    Hid st(H5Tcopy(H5T_C_S1), H5Tclose);
    H5Tset_size(st, H5T_VARIABLE);
    H5Tset_cset(st, H5T_CSET_UTF8);
    int32_t rc = write_attr(w->group, "scenario_id", st, st, -1, &scenario_id);
    if (!rc) rc = write_attr(w->group, "scenario_center", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, n_center, center);
    if (!rc) rc = write_attr(w->group, "scenario_yaw", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, -1, &yaw);
    const int8_t wm = with_map != 0;
    if (!rc) rc = write_attr(w->group, "with_map", w->bool_type, w->bool_type, -1, &wm);
    return rc;
}

int32_t tb_h5_writer_options(tb_h5_writer* w, int32_t deflate_level, int32_t shuffle, int32_t chunk_div) {
    if (!w || deflate_level < 0 || deflate_level > 9 || chunk_div < 0) return fail(TB_H5_ERR_ARG, "tb_h5_writer_options: bad argument");
    w->deflate = deflate_level, w->shuffle = shuffle != 0, w->chunk_div = chunk_div;
    return 0;
}

int32_t tb_h5_writer_dataset(tb_h5_writer* w, const char* key, int32_t kind, const int64_t* dims, int32_t rank, const void* data) {
    if (!w || !key || !data || (rank > 0 && !dims) || rank < 0 || rank > 8) return fail(TB_H5_ERR_ARG, "tb_h5_writer_dataset: bad argument");
    if (w->group < 0) return fail(TB_H5_ERR_ARG, "tb_h5_writer_dataset: no episode started");
    hid_t ftype, mtype;
    if (kind == TB_H5_F32) ftype = H5T_IEEE_F32LE, mtype = H5T_NATIVE_FLOAT;
    else if (kind == TB_H5_MASK_U8) ftype = mtype = w->bool_type;
    else if (kind == TB_H5_I64) ftype = H5T_STD_I64LE, mtype = H5T_NATIVE_INT64;
    else return fail(TB_H5_ERR_ARG, "tb_h5_writer_dataset: kind must be F32, MASK_U8 or I64");
    hsize_t hd[8];
    for (int i = 0; i < rank; ++i) hd[i] = (hsize_t)dims[i];
    Hid sp(rank ? H5Screate_simple(rank, hd, nullptr) : H5Screate(H5S_SCALAR), H5Sclose);
    Hid dcpl(H5Pcreate(H5P_DATASET_CREATE), H5Pclose);
    if (rank > 0 && count_of(dims, rank) > 0 && w->chunk_div > 0) {  // readers do not see the chunk shape
        hsize_t cd[8];
        for (int i = 0; i < rank; ++i) cd[i] = (hd[i] + (hsize_t)w->chunk_div - 1) / (hsize_t)w->chunk_div;
        H5Pset_chunk(dcpl, rank, cd);
        if (w->shuffle) H5Pset_shuffle(dcpl);
        if (w->deflate > 0) H5Pset_deflate(dcpl, (unsigned)w->deflate);
    }
    Hid d(H5Dcreate2(w->group, key, ftype, sp, w->lcpl, dcpl, H5P_DEFAULT), H5Dclose);
    if (!d.ok()) return fail(TB_H5_ERR_IO, std::string("tb_h5_writer_dataset: cannot create ") + key);
    if (count_of(dims, rank) > 0 && H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0)
        return fail(TB_H5_ERR_IO, std::string("tb_h5_writer_dataset: cannot write ") + key);
    return 0;
}

int32_t tb_h5_writer_close(tb_h5_writer* w, int64_t data_len) {
    if (!w) return fail(TB_H5_ERR_ARG, "tb_h5_writer_close: null handle");
    int32_t rc = write_attr(w->file, "data_len", H5T_STD_I64LE, H5T_NATIVE_INT64, -1, &data_len);
    if (w->group >= 0) H5Gclose(w->group);
    H5Tclose(w->bool_type);
    H5Pclose(w->lcpl);
    if (H5Fclose(w->file) < 0 && !rc) rc = fail(TB_H5_ERR_IO, "tb_h5_writer_close: close failed");
    delete w;
    return rc;
}

}  // extern "C"
