// Adam on one Gaussian (K18) and its fp16 re-pack (K19): shared by optimizer.hip and by the kernel that runs them straight after the geometry
// backward of the same Gaussian (backward.hip).  See optimizer.hip for the layout notes.
#pragma once
#include "common.h"
#include "wgslm.h"

struct Adam3 { float p, m, v; };
WD_DEV Adam3 adam_step(const wdgs_adam_hyperparameters& h, float param, float grad, float m, float v, float lr) {
    const float m_new = h.beta1 * m + (1.0f - h.beta1) * grad;
    const float v_new = h.beta2 * v + (1.0f - h.beta2) * grad * grad;
    const float step = wd_div(-lr * m_new, wd_sqrt(v_new) + h.epsilon);
    return Adam3{param + step, m_new, v_new};
}

struct Grad14 { float pos[3], opac, rot[4], scale[3], color[3]; };

WD_DEV Grad14 unpack_gradient(const u32* __restrict__ gradients, u32 idx) {
    const uint4* gp = reinterpret_cast<const uint4*>(gradients + (size_t)idx * 8);
    const uint4 a = gp[0], b = gp[1];
    Grad14 g;
    g.pos[0] = wd_unpack_lo(a.x); g.pos[1] = wd_unpack_hi(a.x); g.pos[2] = wd_unpack_lo(a.y); g.opac = wd_unpack_hi(a.y);
    g.rot[0] = wd_unpack_lo(a.z); g.rot[1] = wd_unpack_hi(a.z); g.rot[2] = wd_unpack_lo(a.w); g.rot[3] = wd_unpack_hi(a.w);
    g.scale[0] = wd_unpack_lo(b.x); g.scale[1] = wd_unpack_hi(b.x); g.scale[2] = wd_unpack_lo(b.y);
    g.color[0] = wd_unpack_lo(b.z); g.color[1] = wd_unpack_hi(b.z); g.color[2] = wd_unpack_lo(b.w);
    return g;
}

// Compact training copy "cs": what the reference spreads over three arrays with padding or long strides -- position {param, m, v} (OptVec4:
// a fourth, unused lane each), log-scale {param, m, v} (same), SH-DC {param, m, v} (192- / 384-byte strides in paramSH / stateSH) -- as 27
// floats per Gaussian in SEVEN PLANES of float4: plane k holds quad k of every Gaussian, planes[k * pitch + idx], so that the 64 lanes of a
// wave read 1 KB of consecutive bytes per plane.  (As one 112-byte row per Gaussian -- round 3's first form -- every load instruction
// touched 7 KB for 1 KB of data: scripts/microbench/hbm_stream.hip measures 4.5-4.7 TB/s for that shape against 6.1 for consecutive
// 16-byte elements, profiles/r04j_hbm_stream.txt.)  The quads of a Gaussian, in plane order:
//   [0-2] pos p  [3-5] pos m  [6-8] pos v  [9-11] scale p  [12-14] scale m  [15-17] scale v  [18-20] dc p  [21-23] dc m  [24-26] dc v  [27] pad
// The reference-layout arrays are brought up to date at every hand-over (cs_flush: get_state / release_state / destroy) and loaded when
// state is adopted, unpacked or rewritten from outside (cs_load).  Rotation (no padding) and opacity stay where they are.
// (CsView: common.h)

// Adam on one Gaussian's 14 trained scalars (SH: DC only) followed by the fp16 re-pack of that Gaussian.
// rows_out (nullable): the re-packed row -- 6 Gaussian words, SH word 0, low half of SH word 1 -- also goes to rows_out[idx*8 ..],
// the 32-byte form in which a data-parallel rank publishes the Gaussians it owns (wdgs_comm_allgather_rows).
WD_DEV void adam_and_repack(u32 idx, bool update, const Grad14& g, const wdgs_adam_hyperparameters& h, float4* __restrict__ opt_rot,
                            float* __restrict__ opt_opacity, const CsView cs, u32* __restrict__ gaussians, u32* __restrict__ sh_buffer,
                            u32* __restrict__ rows_out = nullptr, u32* __restrict__ dc_words = nullptr) {
    const float4 q0 = cs.quad(0, idx), q2 = cs.quad(2, idx), q4 = cs.quad(4, idx), q5 = cs.quad(5, idx);
    float4 R = opt_rot[(size_t)idx * 3];
    float op = opt_opacity[(size_t)idx * 3];
    float Px = q0.x, Py = q0.y, Pz = q0.z;          // position
    float Sx = q2.y, Sy = q2.z, Sz = q2.w;          // log-scale
    float c0 = q4.z, c1 = q4.w, c2 = q5.x;          // SH DC
    if (update) {
        const float4 q1 = cs.quad(1, idx), q3 = cs.quad(3, idx), q6 = cs.quad(6, idx);
        const Adam3 px = adam_step(h, Px, g.pos[0], q0.w, q1.z, h.lr_pos), py = adam_step(h, Py, g.pos[1], q1.x, q1.w, h.lr_pos),
                    pz = adam_step(h, Pz, g.pos[2], q1.y, q2.x, h.lr_pos);
        Px = px.p; Py = py.p; Pz = pz.p;
        {
            const float4 m = opt_rot[(size_t)idx * 3 + 1], v = opt_rot[(size_t)idx * 3 + 2];
            const Adam3 rx = adam_step(h, R.x, g.rot[0], m.x, v.x, h.lr_rot), ry = adam_step(h, R.y, g.rot[1], m.y, v.y, h.lr_rot),
                        rz = adam_step(h, R.z, g.rot[2], m.z, v.z, h.lr_rot), rw = adam_step(h, R.w, g.rot[3], m.w, v.w, h.lr_rot);
            const vec4 nr = normalize(V4(rx.p, ry.p, rz.p, rw.p));
            R = make_float4(nr.x, nr.y, nr.z, nr.w);
            opt_rot[(size_t)idx * 3] = R;
            opt_rot[(size_t)idx * 3 + 1] = make_float4(rx.m, ry.m, rz.m, rw.m);
            opt_rot[(size_t)idx * 3 + 2] = make_float4(rx.v, ry.v, rz.v, rw.v);
        }
        const Adam3 sx = adam_step(h, Sx, g.scale[0], q3.x, q3.w, h.lr_scale), sy = adam_step(h, Sy, g.scale[1], q3.y, q4.x, h.lr_scale),
                    sz = adam_step(h, Sz, g.scale[2], q3.z, q4.y, h.lr_scale);
        Sx = sx.p; Sy = sy.p; Sz = sz.p;
        {
            const Adam3 r = adam_step(h, op, g.opac, opt_opacity[(size_t)idx * 3 + 1], opt_opacity[(size_t)idx * 3 + 2], h.lr_opacity);
            op = r.p;
            opt_opacity[(size_t)idx * 3] = r.p;
            opt_opacity[(size_t)idx * 3 + 1] = r.m;
            opt_opacity[(size_t)idx * 3 + 2] = r.v;
        }
        const Adam3 r0 = adam_step(h, c0, g.color[0], q5.y, q6.x, h.lr_color), r1 = adam_step(h, c1, g.color[1], q5.z, q6.y, h.lr_color),
                    r2 = adam_step(h, c2, g.color[2], q5.w, q6.z, h.lr_color);
        c0 = r0.p; c1 = r1.p; c2 = r2.p;
        cs.quad(0, idx) = make_float4(px.p, py.p, pz.p, px.m);
        cs.quad(1, idx) = make_float4(py.m, pz.m, px.v, py.v);
        cs.quad(2, idx) = make_float4(pz.v, sx.p, sy.p, sz.p);
        cs.quad(3, idx) = make_float4(sx.m, sy.m, sz.m, sx.v);
        cs.quad(4, idx) = make_float4(sy.v, sz.v, r0.p, r1.p);
        cs.quad(5, idx) = make_float4(r2.p, r0.m, r1.m, r2.m);
        cs.quad(6, idx) = make_float4(r0.v, r1.v, r2.v, 0.0f);
    }
    struct { float x, y, z; } P = {Px, Py, Pz}, S = {Sx, Sy, Sz};
    // re-pack (update-gaussians.wgsl:41-75): whole Gaussian, SH word 0, low half of SH word 1
    u32* gp = gaussians + (size_t)idx * 6;
    *reinterpret_cast<uint2*>(gp) = make_uint2(wd_pack2(P.x, P.y), wd_pack2(P.z, op));
    *reinterpret_cast<uint2*>(gp + 2) = make_uint2(wd_pack2(R.x, R.y), wd_pack2(R.z, R.w));
    *reinterpret_cast<uint2*>(gp + 4) = make_uint2(wd_pack2(S.x, S.y), wd_pack2(S.z, 0.0f));
    // SH word 0 and the LOW half of word 1 (the third DC coefficient) as a 4-byte and a 2-byte store.  The reference writes
    // pack2x16float(c2, unpack2x16float(old).y): the neighbouring coefficient survives that round trip bit for bit (only a NaN
    // payload could change, which WGSL leaves implementation-defined), so not touching it is the same result -- without fetching
    // the 96-byte row's cache line just to copy 2 bytes back (it was 19 % of this kernel's HBM traffic).
    const u32 sh0 = wd_pack2(c0, c1), sh1lo = wd_f16bits(c2) & 0xFFFFu;
    if (dc_words) {
        // Deferred SH writes (wdgs_optimizer_set_deferred_sh): the three trained halves go to a compact u32[N][2] array that project_count
        // reads in place of the row's first six bytes -- an 8-byte store, 512 contiguous bytes per wave -- and the 96-byte rows are
        // brought up to date at hand-over points (wdgs_optimizer_flush_sh).  The 6-byte store below is a partial write of a 64-byte
        // memory word: ~120 MB of read-modify-write per step at 1 M Gaussians (profiles/r02o_hbm_by_kernel.md), a fifth of this kernel's traffic.
        *reinterpret_cast<uint2*>(dc_words + (size_t)idx * 2) = make_uint2(sh0, sh1lo);
    } else {
        u32* shp = sh_buffer + (size_t)idx * 24;
        shp[0] = sh0;
        reinterpret_cast<unsigned short*>(shp)[2] = (unsigned short)sh1lo;
    }
    if (rows_out) {
        uint4* ro = reinterpret_cast<uint4*>(rows_out + (size_t)idx * 8);
        ro[0] = make_uint4(wd_pack2(P.x, P.y), wd_pack2(P.z, op), wd_pack2(R.x, R.y), wd_pack2(R.z, R.w));
        ro[1] = make_uint4(wd_pack2(S.x, S.y), wd_pack2(S.z, 0.0f), sh0, sh1lo);
    }
}
