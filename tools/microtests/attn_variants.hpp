// Candidate variants of attention_head_x (tb_device_xdl.hpp) for tools/microtests/attn_loop.hip.  Same operand layouts and the
// same arithmetic per key; they differ in how many keys share one step of the online softmax and in the prefetch distance.
#pragma once

namespace tb {
namespace TB_XNS {

// ---- 64 keys (two 32-key blocks) per online-softmax step ---------------------------------------------------------------
struct KFragX2 {
    xh8 ka[4][NPL];  // [key tile of the pair: block a tiles 0,1, block b tiles 2,3][plane]
    f32x4 kb[4];
};
struct VFragX2 {
    xh8 va[2][2][NPL];  // [block][d tile][plane]
};

__device__ __forceinline__ void k_load_x2(KFragX2& f, const xhalf* __restrict__ kfb, const float* __restrict__ bbase, int ka0, int kb0) {
    const xhalf* pa = kfb + (size_t)(ka0 >> 5) * KV_BLOCK_HALFS;
    const xhalf* pb = kfb + (size_t)(kb0 >> 5) * KV_BLOCK_HALFS;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            f.ka[t][pl] = *reinterpret_cast<const xh8*>(pa + (pl * 2 + t) * 512);
            f.ka[2 + t][pl] = *reinterpret_cast<const xh8*>(pb + (pl * 2 + t) * 512);
        }
        f.kb[t] = ldg4(bbase + ka0 + 16 * t);
        f.kb[2 + t] = ldg4(bbase + kb0 + 16 * t);
    }
}
__device__ __forceinline__ void v_load_x2(VFragX2& f, const xhalf* __restrict__ vfb, int ka0, int kb0) {
    const xhalf* pa = vfb + (size_t)(ka0 >> 5) * KV_BLOCK_HALFS;
    const xhalf* pb = vfb + (size_t)(kb0 >> 5) * KV_BLOCK_HALFS;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            f.va[0][dt][pl] = *reinterpret_cast<const xh8*>(pa + (pl * 2 + dt) * 512);
            f.va[1][dt][pl] = *reinterpret_cast<const xh8*>(pb + (pl * 2 + dt) * 512);
        }
}

__device__ __forceinline__ void attn_qk_x2(const KFragX2& f, const xh8& qh, const xh8& ql, f32x4 (&s)[4], f32x4 (&c)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        c[t] = splat(0.f);
        if (NPL == 2) c[t] = mfma_h(f.ka[t][0], ql, c[t]);
        s[t] = mfma_h(f.ka[t][0], qh, splat(0.f));
        if (NPL == 2) c[t] = mfma_h(f.ka[t][P1], qh, c[t]);
    }
}

// logits (log2 units) of the 16 keys of this lane, their max over the wave's four row groups, the new running max and alpha.
// `b_off` = true masks block b (odd tail: the pair's second block does not exist)
__device__ __forceinline__ void attn_stats_x2(const f32x4 (&s)[4], const f32x4 (&c)[4], const f32x4 (&kb)[4], bool b_off, float run_max,
                                              float (&sv)[16], float& new_max, float& alpha) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = fmaf(s[t][r] + c[t][r] * SPLIT_INV, ATTN_SCALE * 1.44269504088896340736f, kb[t][r]);
            sv[t * 4 + r] = (t >= 2 && b_off) ? -INFINITY : v;
        }
    float m0 = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    float m1 = fmaxf(fmaxf(fmaxf(sv[8], sv[9]), fmaxf(sv[10], sv[11])), fmaxf(fmaxf(sv[12], sv[13]), fmaxf(sv[14], sv[15])));
    const float tmax = rows_max(fmaxf(m0, m1));
    new_max = fmaxf(run_max, tmax);
    alpha = exp2_neg(run_max - new_max);
}

template <int VAR>
__device__ __forceinline__ bool attention_head_var(const f32x4 (&q)[2], AttnPreX& pre, const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                   const float* __restrict__ keybias, int n_key_pad, int kstart, int head, int lane,
                                                   f32x4 (&o)[2], WUnitX& un, const WNextX& nx) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    xh8 qh, ql;
    split8(q[0], q[1], qh, ql);
    f32x4 oh[2] = {splat(0.f), splat(0.f)}, oc[2] = {splat(0.f), splat(0.f)};
    const int nblk = n_key_pad >> 5;
    const int npair = (nblk + 1) >> 1;
    const bool odd = (nblk & 1) != 0;
    // pair j covers blocks 2j, 2j+1 of the (wrapped) walk that starts at kstart
    auto blk = [&](int i) { return kwrap(kstart + 32 * (i < nblk ? i : nblk - 1), n_key_pad); };  // (clamped re-reads past the end)
    KFragX2 kn;
    VFragX2 vc;
    float run_max = -INFINITY, run_sum = 0.f, new_max, alpha, sv[16];
    TB_SCHED_FENCE();
    {
        KFragX2 k0;
        k_load_x2(k0, kbase, bbase, blk(0), blk(1));
        v_load_x2(vc, vbase, blk(0), blk(1));
        k_load_x2(kn, kbase, bbase, blk(2), blk(3));
        TB_SCHED_FENCE();
        f32x4 s[4], c[4];
        attn_qk_x2(k0, qh, ql, s, c);
        attn_stats_x2(s, c, k0.kb, odd && npair == 1, run_max, sv, new_max, alpha);
    }
    const int j_issue = npair >= 2 ? npair - 2 : 0;
    for (int j = 0; j < npair; ++j) {
        TB_SCHED_FENCE();
        // QK of the next pair (XDL) under the exponentials of this one
        f32x4 ts[4], tc[4];
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_qk_x2(kn, qh, ql, ts, tc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            in_vgpr(ts[t]);
            in_vgpr(tc[t]);
        }
        const f32x4 nb[4] = {kn.kb[0], kn.kb[1], kn.kb[2], kn.kb[3]};
        TB_SCHED_FENCE();
        k_load_x2(kn, kbase, bbase, blk(2 * j + 4), blk(2 * j + 5));
        if (j == j_issue) wloadx(un, nx, lane);
        float p[16];
        const bool skip = (VAR == 3) && __builtin_amdgcn_readfirstlane(__float_as_uint(alpha)) == 0x3f800000u && __all(alpha == 1.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = exp2_neg(sv[r] - new_max);
        const float psum = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) +
                           (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
        run_sum = run_sum * alpha + psum;
        run_max = new_max;
        xh8 ph[2], pl[2];
        split8<false>(f32x4{p[0], p[1], p[2], p[3]}, f32x4{p[4], p[5], p[6], p[7]}, ph[0], pl[0]);
        split8<false>(f32x4{p[8], p[9], p[10], p[11]}, f32x4{p[12], p[13], p[14], p[15]}, ph[1], pl[1]);
        if (!skip) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                oh[dt] *= splat(alpha);
                oc[dt] *= splat(alpha);
            }
        }
        TB_SCHED_FENCE();
        // PV of this pair (XDL) under the scale / mask / running max of the next
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (NPL == 2) oc[dt] = mfma_h(vc.va[b][dt][0], pl[b], oc[dt]);
                oh[dt] = mfma_h(vc.va[b][dt][0], ph[b], oh[dt]);
                if (NPL == 2) oc[dt] = mfma_h(vc.va[b][dt][P1], ph[b], oc[dt]);
            }
        TB_SCHED_FENCE();
        v_load_x2(vc, vbase, blk(2 * j + 2), blk(2 * j + 3));
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_stats_x2(ts, tc, nb, odd && (j + 2 == npair), run_max, sv, new_max, alpha);  // (unused after the last pair)
        TB_SCHED_FENCE();
    }
    run_sum = rows_sum(run_sum);
    const bool novalid = !(run_sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / run_sum;
    o[0] = (oh[0] + oc[0] * splat(SPLIT_INV)) * splat(inv);
    o[1] = (oh[1] + oc[1] * splat(SPLIT_INV)) * splat(inv);
    return novalid;
}


// ---- VAR 4: TWO independent online-softmax chains per wave (chain c walks blocks c * nblk/2 ...), interleaved in one instruction
// stream and merged at the end: the instruction-level twin of two waves per SIMD.  nblk must be even and >= 4.
__device__ __forceinline__ bool attention_head_dual(const f32x4 (&q)[2], const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                    const float* __restrict__ keybias, int n_key_pad, int kstart, int head, int lane,
                                                    f32x4 (&o)[2], WUnitX& un, const WNextX& nx) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    xh8 qh, ql;
    split8(q[0], q[1], qh, ql);
    const int nblk = n_key_pad >> 5, nh = nblk >> 1;
    f32x4 oh[2][2], oc[2][2];
    KFragX kn[2];
    VFragX vc[2];
    float run_max[2], run_sum[2], new_max[2], alpha[2], sv[2][8];
    int kc[2], k1[2], k2[2];
    TB_SCHED_FENCE();
    {
        KFragX k0[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            kc[c] = kwrap(kstart + c * nh * 32, n_key_pad);
            k1[c] = kwrap(kc[c] + 32, n_key_pad);
            k2[c] = kwrap(k1[c] + 32, n_key_pad);
            k_load_x(k0[c], kbase, bbase, kc[c]);
            v_load_x(vc[c], vbase, kc[c]);
            k_load_x(kn[c], kbase, bbase, k1[c]);
            oh[c][0] = oh[c][1] = oc[c][0] = oc[c][1] = splat(0.f);
            run_max[c] = -INFINITY;
            run_sum[c] = 0.f;
        }
        TB_SCHED_FENCE();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 s[2], cc[2];
            attn_qk_x(k0[c], qh, ql, s, cc);
            attn_stats_x<false>(s, cc, k0[c].kb, 0, -1, run_max[c], sv[c], new_max[c], alpha[c]);
        }
    }
    const int i_issue = nh >= 2 ? nh - 2 : 0;
    for (int i = 0; i < nh; ++i) {
        int kn1[2], kld[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            kn1[c] = (i + 1 < nh) ? k1[c] : kc[c];
            kld[c] = (i + 2 < nh) ? k2[c] : kc[c];
        }
        TB_SCHED_FENCE();
        f32x4 ts[2][2], tc[2][2], nb[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            attn_qk_x(kn[c], qh, ql, ts[c], tc[c]);
            nb[c][0] = kn[c].kb[0];
            nb[c][1] = kn[c].kb[1];
        }
        TB_SCHED_FENCE();
        xh8 ph[2], pl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            k_load_x(kn[c], kbase, bbase, kld[c]);
            float p[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) p[r] = exp2_neg(sv[c][r] - new_max[c]);
            run_sum[c] = run_sum[c] * alpha[c] + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
            run_max[c] = new_max[c];
            split8<false>(f32x4{p[0], p[1], p[2], p[3]}, f32x4{p[4], p[5], p[6], p[7]}, ph[c], pl[c]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                oh[c][dt] *= splat(alpha[c]);
                oc[c][dt] *= splat(alpha[c]);
            }
        }
        if (i == i_issue) wloadx(un, nx, lane);
        TB_SCHED_FENCE();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (NPL == 2) oc[c][dt] = mfma_h(vc[c].va[dt][0], pl[c], oc[c][dt]);
                oh[c][dt] = mfma_h(vc[c].va[dt][0], ph[c], oh[c][dt]);
                if (NPL == 2) oc[c][dt] = mfma_h(vc[c].va[dt][P1], ph[c], oc[c][dt]);
            }
        TB_SCHED_FENCE();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            v_load_x(vc[c], vbase, kn1[c]);
            attn_stats_x<false>(ts[c], tc[c], nb[c], 0, -1, run_max[c], sv[c], new_max[c], alpha[c]);
            kc[c] = k1[c];
            k1[c] = k2[c];
            k2[c] = kwrap(k2[c] + 32, n_key_pad);
        }
        TB_SCHED_FENCE();
    }
    const float m = fmaxf(run_max[0], run_max[1]);
    const float f0 = exp2_neg(run_max[0] - m), f1 = exp2_neg(run_max[1] - m);
    const float rs = rows_sum(run_sum[0] * f0 + run_sum[1] * f1);
    const bool novalid = !(rs > 0.f);
    const float inv = novalid ? 0.f : 1.0f / rs;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
        o[dt] = ((oh[0][dt] + oc[0][dt] * splat(SPLIT_INV)) * splat(f0) + (oh[1][dt] + oc[1][dt] * splat(SPLIT_INV)) * splat(f1)) * splat(inv);
    return novalid;
}

}  // namespace TB_XNS
}  // namespace tb
