// sa_fused.hip -- fused grouped-MLP + max of one PointNet++ set-abstraction scale (eval mode).
//
// The reference materialises the grouped tensor (B, Cin, S, K) with advanced indexing and then
// runs [Conv2d 1x1 + BatchNorm2d + ReLU] x3 and a max over K as ~12 separate kernels
// (pointnet_utils.py:389-403, :566-581).  On MI355X that is pure HBM traffic over tensors that
// exist only to be reduced.  Here one kernel consumes the neighbour indices directly:
//
//   layer 1 is linear in [feat_j | xyz_j - c_s | centre_feat_s], so its per-POINT half A1[b,j,:] is a dense GEMM over
//   the N points done by the caller (library GEMM); in the kernel layer 1 is
//       h1 = relu(A1[idx] + Wx (xyz[idx] - c_s) + b1 + cadd_s)
//   -- a coalesced row gather + a few packed FMAs, written to LDS by the LOAD role;
//   layers 2 and 3 run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 157 TF
//   peak) with the BN-folded weights held in REGISTERS as the B operand for the whole kernel
//   (the workgroup is persistent over position tiles), activations staged through LDS as the
//   A operand; bias is the accumulator's initial value, ReLU is applied on the way to LDS;
//   the max over the K neighbours is taken on the accumulator registers (rows of the MFMA
//   D tile are positions) + two cross-lane steps, so (B, C, S, K) never exists.
//
// Tile geometry: 4 COMPUTE waves = WC channel groups x WP position groups (WC*WP = 4); a position
// group owns 64 consecutive (s,k) positions = 64/K centroids (K in {16,32,64}); 4 LOAD waves prepare the next tile.
//
// What bounds it (profiles/r02_misc_measurements.md): on this part the fp32 matrix rate equals the packed-fp32 vector
// rate, and next to a wave that streams fp32 MFMAs the SIMD issues almost nothing else -- VALU work does not hide under
// the MFMAs, wherever it is placed and whatever the wave priorities.  A tile costs its 640 MFMAs x 32 cycles plus about
// four cycles per non-MFMA instruction of either role plus the barrier / LDS latencies, so both roles are written for
// instruction count: buffer-descriptor gathers, no per-row clamps or divisions, packed FMAs, one store per centroid.
#include <stdlib.h>

#include "pn2_common.h"
#include "../../include/pn2_ext.h"

#ifndef SA_PAD
#define SA_PAD 4  /* LDS row padding in floats (row stride C + SA_PAD); must keep rows 16-byte aligned */
#endif

// keeps the prefetching ds_reads of the MFMA loops where they are written (the scheduler otherwise sinks them back to
// their use); -DSA_PREFETCH=0 restores the round-1 order for A/B measurements
#ifndef SA_DEFER_EPILOGUE
/* 1 = a tile's max-pool + store runs after the first k-step of the NEXT tile's layer 2 instead of between the last MFMA of
   layer 3 and the barrier.  Measured SLOWER (pair launch 76.5 -> 77.6 us): VALU work placed inside the MFMA stream is not
   free on this part, it costs the matrix pipe more than the same instructions cost in the bubble. */
#define SA_DEFER_EPILOGUE 0
#endif
#ifndef SA_OVERLAP_WRITEOUT
#define SA_OVERLAP_WRITEOUT 1  /* layer 2: ReLU -> H2 write-out of pass p after the first k-step of pass p+1 (second accumulator set) */
#endif
#ifndef SA_SCALAR_L1
#define SA_SCALAR_L1 0  /* layer-1 arithmetic of the LOAD role with scalar fp32 ops instead of packed ones (A/B: scripts/probes/sa_sweep_kernel.sh) */
#endif
#ifndef SA_LOADER_PRIO
#define SA_LOADER_PRIO 0
#endif
#ifndef SA_PROBE
#define SA_PROBE 0  /* timing probes of the LOAD role (scripts/probes): 1 = no row gathers, 2 = no layer-1 arithmetic / LDS writes, 4 = no index loads */
#endif
#ifndef SA_PREFETCH
#define SA_PREFETCH 1
#endif
#if SA_PREFETCH
#define SA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SA_SCHED_FENCE() ((void)0)
#endif

namespace pn2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
// 12-byte xyz record read as ONE global_load_dwordx3 (dword alignment is enough): next to a running MFMA
// stream every vector-memory instruction costs the issuing wave ~100-185 cycles, so the LOAD role counts them.
__device__ __forceinline__ f32x3 load_xyz(const float *p) {
    f32x3 v;
    __builtin_memcpy(&v, p, 12);
    return v;
}

// Buffer descriptor (SRSRC) loads: wave-uniform base in SGPRs + one 32-bit per-lane byte offset, hardware bounds check
// (reads past `bytes` return 0).  Built from scalars only, so hipcc keeps the descriptor in SGPRs (no waterfall loop).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
// (whole-vector bit casts: __builtin_bit_cast(float, v.x) on a vector ELEMENT reads element 0 whatever the subscript
// with this hipcc -- clang takes the address of the vector for the element lvalue.)
// a * b + c on the 24-bit integer multiplier (full rate; v_mul_lo_u32 is quarter rate).  In asm: hipcc's __umul24 is a
// device-library call, which this file's no-IEEE-mode functions (see _build.py) cannot inline.
__device__ __forceinline__ unsigned mad24(int a, unsigned b_uniform, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mul24(int a, unsigned b_uniform) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b_uniform), "v"(a));
    return r;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
// voff: per-lane byte offset (bounds-checked against the descriptor), soff: wave-uniform byte offset added to the base
__device__ __forceinline__ float4 ld_b128(rsrc_t r, unsigned voff, int soff) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
// acc + w * d.lo / acc + w * d.hi on two channels at once (v_pk_fma_f32 with the broadcast chosen by op_sel).  Written in
// asm so that `d` is a REAL register pair: when hipcc forms the broadcast itself it pairs the one live value with whatever
// register follows it, and if that neighbour is the destination of a load still in flight the hazard tracker waits for
// the load -- the loader's prefetches were drained by an s_waitcnt vmcnt(0) in front of the first FMA of every half tile.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 w, f32x2 d, f32x2 acc) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w), "v"(d), "v"(acc));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 w, f32x2 d, f32x2 acc) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(w), "v"(d), "v"(acc));
    return r;
}

struct SaArgs {
    int B, N, S, K, lgK;
    const float *a1f;   // (B,N,a1f_ld>=C1) point-major per-point feature term of layer 1, or nullptr
    const float *xyz;   // (B,N,3) point coordinates, or nullptr (no relative-coordinate term)
    const float *cxyz;  // (B,S,3) centroid coordinates (with xyz)
    const float *wx;    // (C1,3) layer-1 weights of the [xyz_j - c_s] channels (with xyz)
    const float *b1;    // (C1) layer-1 bias, or nullptr
    const float *cadd;  // (B,S,cadd_ld>=C1) per-centroid additive term (centre features), or nullptr
    int a1f_ld, cadd_ld;
    const int *idx;     // (B,S,K)
    const float *w2, *b2, *w3, *b3;
    const float *w2e;   // (C2,3) weights of the three extra input columns of the first MFMA layer (MODE 5), or nullptr
    float *out;         // out[b*out_b + s*out_s + c*out_c]
    long out_b;
    int out_s, out_c;
    int num_tiles, tiles_per_cloud;
    long long *trace;   // debug: per-phase s_memtime stamps of workgroup 0 (nullptr = off)
};

// Workgroup = 8 waves with fixed roles (wave specialisation):
//   waves 0-3  COMPUTE: layers 2 and 3 on the matrix cores, weights register-resident, max over K, store;
//   waves 4-7  LOAD:    gather + layer-1 arithmetic of the NEXT tile into the other slot of a double-buffered
//                       LDS tile (software-pipelined: indices two half tiles ahead, rows one), so the matrix
//                       cores never wait for a memory round trip.
// Two s_barriers per tile keep the roles in step (H1[next] complete / H2 reusable).
// RTC = row tiles (of 16 positions) whose accumulators are live at once (4 = fewest passes over the weight
// registers, 2 = half the accumulator / A-fragment registers).  MINW = waves per SIMD for __launch_bounds__.
// K (neighbours per centroid) is a template parameter: the max-combine / store part is then straight-line code.
// MODE fixes which layer-1 operands exist so the LOAD role is branch-free: 0 = xyz only (sa1), 1 = a1f + xyz,
// 2 = a1f + xyz + cadd, 3 = any combination, tested at run time, 4 = plain rows (two-layer MLP, no pooling).
// The tile loop of one problem, run by workgroup `wg` of the `nwg` workgroups assigned to it (a whole launch, or one share
// of a launch that serves two scales of a module at once -- sa_mlp_max_pair_kernel below).
template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K, int MODE>
__device__ __forceinline__ void sa_body(const SaArgs &A, const int wg, const int nwg) {
    static_assert(K == 16 || K == 32 || K == 64, "K");
    // MODE 4 ("rows"): no neighbourhoods at all -- H1 = the input rows a1f[r, :] as they are, and every row of the layer-3
    // output is stored (no max-pool): a fused two-layer MLP over R = S*K plain rows (pn2x_mlp2_rows), same tile loop.
    // MODE 5: rows whose first layer has three more input columns x[r, C1 .. C1+2] (coordinates next to the features: the
    // [interpolated | xyz] rows of feature propagation): they travel in the padding floats of the LDS rows and their
    // term W2e x_e is added on the VALU when the layer-2 accumulators are written out.
    constexpr bool ROWS = MODE == 4 || MODE == 5, ROWS_EXTRA = MODE == 5;
    static_assert(!ROWS_EXTRA || SA_PAD >= 4, "the extra input columns live in the LDS row padding");
    // MODE 3 runs the code of MODE 2 (all three operands); an absent operand gets a zero-length buffer descriptor, whose loads
    // return 0 through the hardware bounds check (and Wx = 0 without coordinates): no run-time branches in the LOAD role, and the
    // register budget of MODE 2 (tested at run time, the 128-128-192 instance spilled 12-24 registers to scratch memory).
    constexpr bool has_a1f = MODE >= 1, has_xyz = !ROWS, has_cadd = MODE == 2 || MODE == 3;
    const bool p_a1f = MODE == 3 ? A.a1f != nullptr : has_a1f, p_xyz = MODE == 3 ? A.xyz != nullptr : has_xyz;
    const bool p_cadd = MODE == 3 ? A.cadd != nullptr : has_cadd;
    constexpr int lgK = K == 16 ? 4 : K == 32 ? 5 : 6;
    const int N = A.N, S = A.S;
    const float *__restrict__ W2 = A.w2, *__restrict__ b2 = A.b2, *__restrict__ W3 = A.w3, *__restrict__ b3 = A.b3;
    float *__restrict__ out = A.out;
    const int num_tiles = A.num_tiles, tiles_per_cloud = A.tiles_per_cloud;
    constexpr int WP = 4 / WC;
    constexpr int TM = WP * 64;
    constexpr int LD1 = C1 + SA_PAD, LD2 = C2 + SA_PAD;
    constexpr int NT2 = C2 / (16 * WC), NT3 = C3 / (16 * WC);
    static_assert(C2 % (16 * WC) == 0 && C3 % (16 * WC) == 0 && C1 % 16 == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // H1 ring of NB1 tiles: the LOAD role runs NB1-1 tiles ahead of the COMPUTE role, so a slow gather (HBM miss,
    // issue arbitration against the MFMA stream) is absorbed by the ring instead of stalling the matrix cores
    float *H1ring = smem;
    float *H2 = smem + NB1 * TM * LD1;

    const int tid = (int)__builtin_amdgcn_workitem_id_x();  // builtins, not threadIdx / blockIdx: those are device-library calls here (see mad24)
    const int lane = tid & 63;
    const int w = tid >> 6;
    const bool compute = w < 4;  // wave-uniform role
    const int wc = (w & 3) % WC, wp = (w & 3) / WC;
    const int li = lane & 15, g = lane >> 4;
    const int SK = S * K;

    // ---- LOAD role: 4 consecutive layer-1 channels per thread (fixed), rows lt/Q1 + i*(256/Q1) ----------
    constexpr int Q1 = C1 / 4;
    static_assert(256 % Q1 == 0, "a loader thread keeps its channel quad across rows");
    const int lt = tid & 255;
    const int c4 = lt % Q1;
    // layer-1 constants of this loader thread's channel quad (registers; the LOAD role's loop is separate from
    // the COMPUTE role's, so they do not compete with the resident weights)
    float wxr[4][3] = {{0.f}};
    float4 b1r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!compute) {
        if (p_xyz) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) wxr[i][k] = A.wx[(4 * c4 + i) * 3 + k];
        }
        if (A.b1) b1r = *reinterpret_cast<const float4 *>(A.b1 + 4 * c4);
    }

    // h1 = relu(a1f[idx] + Wx (p_idx - c_s) + b1 + cadd_s) for HALF h (rows h*TM/2 ...) of a tile -> LDS, as three
    // stages the LOAD loop software-pipelines: (1) neighbour indices, (2) the row gathers that depend on them,
    // (3) arithmetic + LDS write.  A half tile costs two dependent HBM round trips (index, then rows, ~2-4k cycles
    // each); issued back to back they made the LOAD role as slow as the matrix-core role (trace: 10.9k / 15.7k
    // cycles per half against 9.5k / 15.1k).  In the pipeline the indices of half q+2 and the rows of half q+1 are
    // in flight while half q is finished, so each half exposes at most the tail of one round trip.
    // All loads use clamped (always valid) addresses and no per-row branches.
    constexpr int RPT = (TM / 2) / (256 / Q1);  // rows per loader thread per half tile
    static_assert((TM / 2) % (256 / Q1) == 0, "half tile rows split evenly over the loader threads");
    static_assert(RPT == 1 || RPT == 2 || RPT == 4, "a thread's neighbour indices are ONE 4/8/16-byte load");
    // A thread owns RPT CONSECUTIVE rows (positions): their neighbour indices are one load, and since K is a multiple of
    // RPT they belong to one centroid, whose coordinates / additive term are loaded once per half tile.
    const int row0 = (lt / Q1) * RPT;
    // Tile cursor: tile = wg + i*nwg -> (cloud b, tile t inside the cloud), advanced incrementally (no division in
    // the loops).
    const int cur_db = nwg / tiles_per_cloud, cur_dt = nwg - cur_db * tiles_per_cloud;
    struct Cursor { int tile, b, t; };
    auto cursor_at = [&](int tile) { Cursor c; c.tile = tile; c.b = tile / tiles_per_cloud; c.t = tile - c.b * tiles_per_cloud; return c; };
    auto cursor_next = [&](Cursor c) {
        c.tile += nwg; c.b += cur_db; c.t += cur_dt;
        if (c.t >= tiles_per_cloud) { c.t -= tiles_per_cloud; ++c.b; }
        return c;
    };
    // Every instruction of this role costs the tile about one issue slot (4 cycles): next to a streaming MFMA wave the
    // SIMD hardly issues anything else (neither s_setprio nor fewer / more loads change that: scripts/probes/sa_probe.sh),
    // so the role's instructions execute in the matrix pipe's bubbles and the COMPUTE waves wait for them at the barriers.
    // Hence: gathers through buffer descriptors (a load = ONE v_mad_u32_u24 for its 32-bit offset + the buffer_load,
    // against ~4 VALU ops of 64-bit address arithmetic per flat load), no per-row clamps (positions past S*K in a cloud's
    // last tile read zero through the bounds check of the per-cloud descriptors of idx / cxyz / cadd; a1f and xyz rows
    // are addressed by neighbour index, always inside the cloud, so one descriptor for the whole tensor + a scalar cloud
    // offset does), per-centroid constants formed once per half tile.  Past-the-end tiles are fetched from the last
    // cloud (valid addresses), never written.
    const unsigned c4x16 = 16u * c4, row0x4 = 4u * row0;
    const unsigned a1f_ldb = 4u * (unsigned)A.a1f_ld, cadd_ldb = 4u * (unsigned)A.cadd_ld;
    const unsigned idx_cloud = 4u * (unsigned)SK, a1f_cloud = (unsigned)N * a1f_ldb, xyz_cloud = 12u * (unsigned)N;
    const unsigned cxyz_cloud = 12u * (unsigned)S, cadd_cloud = (unsigned)S * cadd_ldb;
    const rsrc_t ra_all = make_rsrc(A.a1f, p_a1f ? (unsigned)A.B * a1f_cloud : 0u);
    const rsrc_t rx_all = make_rsrc(A.xyz, p_xyz ? (unsigned)A.B * xyz_cloud : 0u);
    struct HalfIdx { int jj[RPT], ss; };
    struct HalfRows { float4 a[RPT], pj[RPT], c, cs, e; };  // pj / cs: (x, y, z, the following record's x -- unused); e: MODE 5
    auto load_idx = [&](const Cursor &cu, int h, HalfIdx &I) {
        const int b = cu.b < A.B ? cu.b : A.B - 1;
        const int pos0 = cu.t * TM + h * (TM / 2);
        if constexpr (ROWS) {  // the "neighbour" of position p is row p
#pragma unroll
            for (int r = 0; r < RPT; ++r) I.jj[r] = pos0 + row0 + r;
            I.ss = 0;
            return;
        }
        const rsrc_t ri = make_rsrc(reinterpret_cast<const char *>(A.idx) + (size_t)b * idx_cloud, idx_cloud);
        if constexpr (RPT == 4) {
            const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)row0x4, 4 * pos0, 0));
            I.jj[0] = v.x; I.jj[1] = v.y; I.jj[2] = v.z; I.jj[3] = v.w;
        } else if constexpr (RPT == 2) {
            const i32x2 v = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, (int)row0x4, 4 * pos0, 0));
            I.jj[0] = v.x; I.jj[1] = v.y;
        } else {
            I.jj[0] = (int)__builtin_amdgcn_raw_buffer_load_b32(ri, (int)row0x4, 4 * pos0, 0);
        }
        I.ss = (pos0 + row0) >> lgK;
    };
    auto load_rows = [&](const Cursor &cu, const HalfIdx &I, HalfRows &D) {
        const int b = cu.b < A.B ? cu.b : A.B - 1;
        if (has_a1f) {
            const int so = ROWS ? 0 : (int)((unsigned)b * a1f_cloud);
#pragma unroll
            for (int r = 0; r < RPT; ++r) D.a[r] = ld_b128(ra_all, mad24(I.jj[r], a1f_ldb, c4x16), so);  // rows past the end read 0
            // MODE 5: the extra columns of the half tile's TM/2 rows, one row per thread of the first loader lanes
            if (ROWS_EXTRA && lt < TM / 2) D.e = ld_b128(ra_all, mad24(I.jj[0] - row0 + lt, a1f_ldb, 4u * C1), so);
        }
        if (has_cadd) {
            const rsrc_t rd = make_rsrc(reinterpret_cast<const char *>(A.cadd) + (size_t)b * cadd_cloud, p_cadd ? cadd_cloud : 0u);
            D.c = ld_b128(rd, mad24(I.ss, cadd_ldb, c4x16), 0);
        }
        if (has_xyz) {
            const int so = (int)((unsigned)b * xyz_cloud);
            const rsrc_t rc = make_rsrc(reinterpret_cast<const char *>(A.cxyz) + (size_t)b * cxyz_cloud, p_xyz ? cxyz_cloud : 0u);
#pragma unroll
            for (int r = 0; r < RPT; ++r) D.pj[r] = ld_b128(rx_all, mul24(I.jj[r], 12u), so);
            D.cs = ld_b128(rc, mul24(I.ss, 12u), 0);
        }
    };
    // The LOAD role shares its SIMD's issue port with a COMPUTE wave: every VALU instruction here can delay an MFMA issue by a
    // slot (layer 2 ran at 36 cycles per MFMA against 32.3 in layer 3, when the loaders mostly wait on memory).  So the
    // layer-1 arithmetic is written with the packed fp32 ops (v_pk_add / v_pk_fma: two channels per instruction)
    // and explicit FMAs, and the per-centroid constants (b1 + cadd) are formed once per group of rows.
    auto finish = [&](const HalfRows &D, float *__restrict__ H1, int h) {
#ifdef SA_FINISH_PRIO
        __builtin_amdgcn_s_setprio(SA_FINISH_PRIO);
#endif
        if constexpr (ROWS) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) *reinterpret_cast<float4 *>(H1 + (h * (TM / 2) + row0 + r) * LD1 + 4 * c4) = D.a[r];
            if (ROWS_EXTRA && lt < TM / 2) *reinterpret_cast<float4 *>(H1 + (h * (TM / 2) + lt) * LD1 + C1) = D.e;
            return;
        }
        const f32x2 w0x = {wxr[0][0], wxr[1][0]}, w0y = {wxr[0][1], wxr[1][1]}, w0z = {wxr[0][2], wxr[1][2]};
        const f32x2 w1x = {wxr[2][0], wxr[3][0]}, w1y = {wxr[2][1], wxr[3][1]}, w1z = {wxr[2][2], wxr[3][2]};
#if SA_SCALAR_L1
        // scalar fp32 form: beside a streaming fp32-MFMA wave a packed fp32 VALU op costs more than the two scalar ops it
        // replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); same IEEE operations, same results
        float t[4] = {b1r.x, b1r.y, b1r.z, b1r.w};
        if (has_cadd) { t[0] += D.c.x; t[1] += D.c.y; t[2] += D.c.z; t[3] += D.c.w; }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float v[4] = {t[0], t[1], t[2], t[3]};
            if (has_a1f) { v[0] += D.a[r].x; v[1] += D.a[r].y; v[2] += D.a[r].z; v[3] += D.a[r].w; }
            if (has_xyz) {
                const float dx = D.pj[r].x - D.cs.x, dy = D.pj[r].y - D.cs.y, dz = D.pj[r].z - D.cs.z;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_fmaf(wxr[i][0], dx, __builtin_fmaf(wxr[i][1], dy, __builtin_fmaf(wxr[i][2], dz, v[i])));
            }
            *reinterpret_cast<float4 *>(H1 + (h * (TM / 2) + row0 + r) * LD1 + 4 * c4) =
                make_float4(fmax_raw(v[0], 0.f), fmax_raw(v[1], 0.f), fmax_raw(v[2], 0.f), fmax_raw(v[3], 0.f));
        }
#else
        // the half tile's centroid: constant term b1 (+ cadd) and coordinates
        f32x2 t0 = {b1r.x, b1r.y}, t1 = {b1r.z, b1r.w};
        if (has_cadd) { t0 += (f32x2){D.c.x, D.c.y}; t1 += (f32x2){D.c.z, D.c.w}; }
        const f32x2 cxy = {D.cs.x, D.cs.y}, czw = {D.cs.z, D.cs.w};
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            f32x2 v0 = t0, v1 = t1;
            if (has_a1f) { v0 += (f32x2){D.a[r].x, D.a[r].y}; v1 += (f32x2){D.a[r].z, D.a[r].w}; }
            if (has_xyz) {
                const f32x2 dxy = (f32x2){D.pj[r].x, D.pj[r].y} - cxy, dzw = (f32x2){D.pj[r].z, D.pj[r].w} - czw;
                v0 = pk_fma_lo(w0x, dxy, pk_fma_hi(w0y, dxy, pk_fma_lo(w0z, dzw, v0)));
                v1 = pk_fma_lo(w1x, dxy, pk_fma_hi(w1y, dxy, pk_fma_lo(w1z, dzw, v1)));
            }
            *reinterpret_cast<float4 *>(H1 + (h * (TM / 2) + row0 + r) * LD1 + 4 * c4) =
                make_float4(fmax_raw(v0.x, 0.f), fmax_raw(v0.y, 0.f), fmax_raw(v1.x, 0.f), fmax_raw(v1.y, 0.f));
        }
#endif
#ifdef SA_FINISH_PRIO
        __builtin_amdgcn_s_setprio(SA_LOADER_PRIO);
#endif
    };
    auto gather = [&](const Cursor &cu, float *__restrict__ H1, int h) {  // unpipelined: prologue only
        HalfIdx I;
        HalfRows D;
        load_idx(cu, h, I);
        load_rows(cu, I, D);
        finish(D, H1, h);
    };

    // ---- COMPUTE role: weights -> registers (B operand: lane holds W[out = tile*16 + li][in = 16*tq + 4*g + j])
    float w2r[NT2][C1 / 4], w3r[NT3][C2 / 4];
    f32x4 bias2v[NT2];  // layer 2 runs with the operands exchanged (see writeout2): a lane owns channels 4 g .. 4 g + 3 of a column tile
    float bias3[NT3];
    f32x4 bias3v[ROWS ? NT3 : 1];  // rows mode: layer 3 runs with the operands swapped (see below), a lane owns 4 consecutive channels
    float w2e[NT2][4][3] = {{{0.f}}};
    if (compute) {
        if constexpr (ROWS_EXTRA) {
#pragma unroll
            for (int ct = 0; ct < NT2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) w2e[ct][r][k] = A.w2e[((wc * NT2 + ct) * 16 + 4 * g + r) * 3 + k];
        }
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) {
            const int oc = (wc * NT2 + ct) * 16 + li;
            {
                const float *bp = b2 + (wc * NT2 + ct) * 16 + 4 * g;
                bias2v[ct] = (f32x4){bp[0], bp[1], bp[2], bp[3]};
            }
#pragma unroll
            for (int tq = 0; tq < C1 / 16; ++tq) {
                const float4 v = *reinterpret_cast<const float4 *>(W2 + (size_t)oc * C1 + 16 * tq + 4 * g);
                w2r[ct][4 * tq + 0] = v.x; w2r[ct][4 * tq + 1] = v.y; w2r[ct][4 * tq + 2] = v.z; w2r[ct][4 * tq + 3] = v.w;
            }
        }
#pragma unroll
        for (int ct = 0; ct < NT3; ++ct) {
            const int oc = (wc * NT3 + ct) * 16 + li;
            bias3[ct] = b3[oc];
            if constexpr (ROWS) {
                const float *bp = b3 + (wc * NT3 + ct) * 16 + 4 * g;
                bias3v[ct] = (f32x4){bp[0], bp[1], bp[2], bp[3]};
            }
#pragma unroll
            for (int tq = 0; tq < C2 / 16; ++tq) {
                const float4 v = *reinterpret_cast<const float4 *>(W3 + (size_t)oc * C2 + 16 * tq + 4 * g);
                w3r[ct][4 * tq + 0] = v.x; w3r[ct][4 * tq + 1] = v.y; w3r[ct][4 * tq + 2] = v.z; w3r[ct][4 * tq + 3] = v.w;
            }
        }
    } else {
        Cursor cu = cursor_at(wg);
        for (int a = 0; a < NB1 - 1; ++a, cu = cursor_next(cu)) {  // prologue: the first NB1-1 tiles of this workgroup
            if (cu.tile < num_tiles) {
                gather(cu, H1ring + a * TM * LD1, 0);
                gather(cu, H1ring + a * TM * LD1, 1);
            }
        }
    }
    __syncthreads();

    // debug trace: stamp(slot) records the shader clock for (workgroup 0, lane 0 of waves 0 and 4)
    // Compiled in only with -DSA_TRACE=1 (scripts/probes/sa_trace.py rebuilds with it): even a never-taken branch per
    // stamp splits the tile body into separate scheduling regions and keeps epilogues from overlapping MFMAs.
    auto stamp = [&](int it, int slot) {
#if defined(SA_TRACE) && SA_TRACE
        if (A.trace && wg == 0 && lane == 0 && (w == 0 || w == 4) && it < 8)
            A.trace[((w >> 2) * 8 + it) * 8 + slot] = (long long)__builtin_readcyclecounter();
#else
        (void)it; (void)slot;
#endif
    };
    // The two roles run SEPARATE loops (same trip count, two s_barriers per tile each), so the register
    // allocator does not have to keep the COMPUTE role's resident weights alive through the LOAD role's code.
    if (!compute) {
        // Wave priority of the LOAD role.  Before its loop was software-pipelined it needed s_setprio(3) to get its
        // (then latency-critical) loads issued between the MFMAs; now its loads run one to two half-tiles ahead and
        // the matrix-core role is the critical path, so default priority is better (sweep, profiles/r01_misc_measurements.md:
        // prio 0 / 1 / 3 -> K=64 launch 71.2 / 72.5 / 72.6 us, sa1 37.8 / 39.6 / 39.7 us).
        __builtin_amdgcn_s_setprio(SA_LOADER_PRIO);
        // pipeline fill: rows of the first half and indices of the second half of the first tile this loop gathers
        HalfIdx I0, I1;
        HalfRows D0, D1;
        Cursor nx = cursor_at(wg + (NB1 - 1) * nwg);  // the tile whose rows this iteration finishes
        Cursor af = cursor_next(nx);                  // the one after it (indices / first rows in flight)
        load_idx(nx, 0, I0);
        load_idx(nx, 1, I1);
        load_rows(nx, I0, D0);
        int it = 0;
        for (int tile = wg; tile < num_tiles; tile += nwg, ++it) {
            stamp(it, 0);
            float *H1n = H1ring + ((it + NB1 - 1) % NB1) * TM * LD1;  // last read by COMPUTE in iteration it-1
#if !(SA_PROBE & 1)
            load_rows(nx, I1, D1);    // half 1 of `nx`: in flight while half 0 is finished
#endif
            stamp(it, 7);
#if !(SA_PROBE & 4)
            load_idx(af, 0, I0);      // indices two halves ahead
#endif
            stamp(it, 1);
#if !(SA_PROBE & 2)
            if (nx.tile < num_tiles) finish(D0, H1n, 0);
#endif
            stamp(it, 2);
            __syncthreads();  // B1
            stamp(it, 3);
#if !(SA_PROBE & 1)
            load_rows(af, I0, D0);    // half 0 of the following tile
#endif
#if !(SA_PROBE & 4)
            load_idx(af, 1, I1);
#endif
            stamp(it, 4);
#if !(SA_PROBE & 2)
            if (nx.tile < num_tiles) finish(D1, H1n, 1);
#endif
            stamp(it, 5);
            __syncthreads();  // B2
            stamp(it, 6);
            nx = af;
            af = cursor_next(af);
        }
        return;
    }
#ifdef SA_COMPUTE_PRIO
    __builtin_amdgcn_s_setprio(SA_COMPUTE_PRIO);
#endif
    int it = 0;
    Cursor cc = cursor_at(wg);
    constexpr int NP = 4 / RTC, NQ2 = C1 / 16, NQ3 = C2 / 16;
    constexpr int rstep = K >> 4;  // row tiles per centroid (K = 16 -> 1, 32 -> 2, 64 -> 4)
    // one store per centroid: lane group g writes channel tile ct = g (the row maxima are in every lane group)
    const int oc_lane = (wc * NT3 + g) * 16 + li;
    const unsigned out_cloud = 4u * ((unsigned)(S - 1) * (unsigned)A.out_s + (unsigned)(C3 - 1) * (unsigned)A.out_c + 1u);
    // Max-pool + store of a tile (optionally deferred into the next tile's layer 2, see SA_DEFER_EPILOGUE; loop-carried then:
    // the row-tile maxima m, the tile's cloud and first position).  The stores go through a buffer descriptor whose bounds
    // check drops the lanes that have nothing to write (offset -1): no exec-mask branch around the store.
    const rsrc_t ro_rows = make_rsrc(out, ROWS ? 4u * (unsigned)N * (unsigned)A.out_s : 0u);  // rows mode: the whole (R = N, ld) output
    float m[4][NT3];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT3; ++ct) m[rt][ct] = 0.f;
    int e_b = 0, e_pos0 = 0;
    bool e_pending = false;
    auto epilogue = [&]() {
        if constexpr (K == 32) {
#pragma unroll
            for (int ct = 0; ct < NT3; ++ct) {
                m[0][ct] = fmax_raw(m[0][ct], m[1][ct]);
                m[2][ct] = fmax_raw(m[2][ct], m[3][ct]);
            }
        }
        if constexpr (K == 64) {
#pragma unroll
            for (int ct = 0; ct < NT3; ++ct) m[0][ct] = fmax_raw(fmax3_raw(m[0][ct], m[1][ct], m[2][ct]), m[3][ct]);
        }
        const rsrc_t ro = make_rsrc(out + (size_t)e_b * A.out_b, e_pending ? out_cloud : 0u);
#pragma unroll
        for (int rt = 0; rt < 4; rt += rstep) {
            const int s = ((e_pos0 + wp * 64) >> lgK) + rt / rstep;
#pragma unroll
            for (int ct0 = 0; ct0 < NT3; ct0 += 4) {
                // the 16 positions of a row tile sit in the 4 lane rows; after rows_max4 every lane row holds the max
                float v = rows_max4(m[rt][ct0]);
#pragma unroll
                for (int j = 1; j < 4; ++j)
                    if (ct0 + j < NT3) {
                        const float vj = rows_max4(m[rt][ct0 + j]);
                        v = g == j ? vj : v;
                    }
                const unsigned off = 4u * ((unsigned)s * (unsigned)A.out_s + (unsigned)(oc_lane + 16 * ct0) * (unsigned)A.out_c);
                const bool ok = ct0 + g < NT3 && s < S;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v, 0.f)), ro, (int)(ok ? off : 0xffffffffu), 0, 0);  // relu commutes with max
            }
        }
    };
    auto frag2 = [&](const float *H1, int p, int tq, int rt) {
        return *reinterpret_cast<const float4 *>(H1 + (wp * 64 + (p * RTC + rt) * 16 + li) * LD1 + 4 * g + 16 * tq);
    };
    // first A fragments of the first tile (later ones are fetched under the last MFMAs of the previous tile's layer 3: half 0
    // of the next H1 tile is complete since that tile's barrier B1)
    // -- possible when the rows of the first pass all lie in half 0: one position group, RTC*16 <= TM/2)
    constexpr bool PREFETCH_NEXT = WP == 1 && RTC * 16 <= TM / 2;
    float4 an2[RTC];
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) an2[rt] = frag2(H1ring, 0, 0, rt);
    for (int tile = wg; tile < num_tiles; tile += nwg, ++it, cc = cursor_next(cc)) {
        stamp(it, 0);
        const float *H1 = H1ring + (it % NB1) * TM * LD1;
        const float *H1next = H1ring + ((it + 1) % NB1) * TM * LD1;
        {
            // ---- layer 2 on the matrix cores ------------------------------------------------------------
            // NP passes of RTC row tiles.  A fragments are fetched ONE k-step ahead, across pass boundaries too (with the
            // ds_reads issued right before their use the matrix pipe idled an LDS round trip per k-step: the only other
            // wave of this SIMD is a LOAD wave, nothing fills it), and a pass's ReLU -> H2 write-out is issued after the
            // first k-step of the NEXT pass (its own accumulator set), i.e. in the shadow of running MFMAs.
            f32x4 acc[2][RTC][NT2];
            // The two operands of the layer-2 instructions trade places (their register layouts are the same: lane (li, g) holds
            // element [li][k = g]), so a 16 x 16 block comes out TRANSPOSED -- D[channel 4 g + r][row li]: a lane owns four
            // consecutive channels of one row and the block goes to H2 as ONE ds_write_b128 per lane instead of four ds_write_b32
            // (conflict-free: LD2 = 4 mod 32, eight lanes cover the 32 banks).  Per-phase stamps of the 32-32-64 instance
            // (scripts/probes/sa1_trace.py): beside the other workgroup's matrix stream this write-out took 2.4 k cycles of an
            // 8.2 k-cycle tile for 16 stores + 16 maxima -- every instruction here waits for an issue slot between two MFMAs.
            auto writeout2 = [&](int p) {  // relu -> H2 (D tile: channel = 4 g + r, row = li)
#pragma unroll
                for (int rt = 0; rt < RTC; ++rt) {
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (ROWS_EXTRA) e = *reinterpret_cast<const float4 *>(H1 + (wp * 64 + (p * RTC + rt) * 16 + li) * LD1 + C1);  // the row's extra input columns
#pragma unroll
                    for (int ct = 0; ct < NT2; ++ct) {
                        float *dst = H2 + (wp * 64 + (p * RTC + rt) * 16 + li) * LD2 + (wc * NT2 + ct) * 16 + 4 * g;
                        f32x4 v = acc[p & 1][rt][ct];
                        if constexpr (ROWS_EXTRA) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                v[r] = __builtin_fmaf(w2e[ct][r][0], e.x, __builtin_fmaf(w2e[ct][r][1], e.y, __builtin_fmaf(w2e[ct][r][2], e.z, v[r])));
                        }
                        *reinterpret_cast<float4 *>(dst) = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    }
                }
            };
            float4 an[RTC];
#pragma unroll
            for (int rt = 0; rt < RTC; ++rt) an[rt] = PREFETCH_NEXT ? an2[rt] : frag2(H1, 0, 0, rt);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
                    for (int ct = 0; ct < NT2; ++ct) acc[p & 1][rt][ct] = bias2v[ct];
#pragma unroll
                for (int tq = 0; tq < NQ2; ++tq) {
                    float4 a[RTC];
#pragma unroll
                    for (int rt = 0; rt < RTC; ++rt) a[rt] = an[rt];
                    if (tq + 1 < NQ2 || p + 1 < NP) {
#pragma unroll
                        for (int rt = 0; rt < RTC; ++rt) an[rt] = tq + 1 < NQ2 ? frag2(H1, p, tq + 1, rt) : frag2(H1, p + 1, 0, rt);
                        SA_SCHED_FENCE();
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rt = 0; rt < RTC; ++rt) {
                            const float av = j == 0 ? a[rt].x : j == 1 ? a[rt].y : j == 2 ? a[rt].z : a[rt].w;
#pragma unroll
                            for (int ct = 0; ct < NT2; ++ct)
                                acc[p & 1][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[ct][4 * tq + j], av, acc[p & 1][rt][ct], 0, 0, 0);
                        }
#if SA_DEFER_EPILOGUE
                    if (!ROWS && tq == 0 && p == 0) epilogue();  // the previous tile's max-pool + store (nothing is written when there was none)
#endif
#if SA_OVERLAP_WRITEOUT
                    if (tq == 0 && p > 0) writeout2(p - 1);
#endif
                }
#if !SA_OVERLAP_WRITEOUT
                writeout2(p);
#endif
            }
            stamp(it, 1);
#if SA_OVERLAP_WRITEOUT
            writeout2(NP - 1);
#endif
        }
        stamp(it, 2);
        __syncthreads();  // B1: H2 complete
        stamp(it, 3);
        {
            // ---- layer 3 + max over the 16 positions of every row tile ------------------------------------------
            auto frag3 = [&](int p, int tq, int rt) {
                return *reinterpret_cast<const float4 *>(H2 + (wp * 64 + (p * RTC + rt) * 16 + li) * LD2 + 4 * g + 16 * tq);
            };
            float4 an[RTC];
#pragma unroll
            for (int rt = 0; rt < RTC; ++rt) an[rt] = frag3(0, 0, rt);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                f32x4 acc[RTC][NT3];
#pragma unroll
                for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
                    for (int ct = 0; ct < NT3; ++ct) acc[rt][ct] = ROWS ? bias3v[ct] : (f32x4){bias3[ct], bias3[ct], bias3[ct], bias3[ct]};
#pragma unroll
                for (int tq = 0; tq < NQ3; ++tq) {
                    float4 a[RTC];
#pragma unroll
                    for (int rt = 0; rt < RTC; ++rt) a[rt] = an[rt];
                    if (tq + 1 < NQ3 || p + 1 < NP) {
#pragma unroll
                        for (int rt = 0; rt < RTC; ++rt) an[rt] = tq + 1 < NQ3 ? frag3(p, tq + 1, rt) : frag3(p + 1, 0, rt);
                    } else if (PREFETCH_NEXT) {  // last k-step of the tile: the first fragments of the NEXT tile's layer 2
#pragma unroll
                        for (int rt = 0; rt < RTC; ++rt) an2[rt] = frag2(H1next, 0, 0, rt);
                    }
                    SA_SCHED_FENCE();
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rt = 0; rt < RTC; ++rt) {
                            const float av = j == 0 ? a[rt].x : j == 1 ? a[rt].y : j == 2 ? a[rt].z : a[rt].w;
#pragma unroll
                            for (int ct = 0; ct < NT3; ++ct) {
                                // rows mode: the two operands trade places (their register layouts are the same: lane (li, g)
                                // holds element [li][k = g]), so the product comes out transposed -- D[channel 4 g + r][row li]:
                                // a lane owns FOUR CONSECUTIVE CHANNELS of one row, and the tile leaves as 16-byte stores
                                if constexpr (ROWS) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[ct][4 * tq + j], av, acc[rt][ct], 0, 0, 0);
                                else acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w3r[ct][4 * tq + j], acc[rt][ct], 0, 0, 0);
                            }
                        }
                }
                if constexpr (ROWS) {  // relu -> out rows; rows past the end are dropped by the bounds check.  One store instruction
                    // per 16 x 16 block instead of four: next to the matrix stream a vector-memory instruction costs the wave
                    // 100+ cycles whatever its width (see the LOAD role), and a 64-row tile of 128 channels was 32 of them per wave
#pragma unroll
                    for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
                        for (int ct = 0; ct < NT3; ++ct) {
                            const int row = cc.t * TM + wp * 64 + (p * RTC + rt) * 16 + li;
                            const unsigned off = 4u * ((unsigned)row * (unsigned)A.out_s + (unsigned)((wc * NT3 + ct) * 16 + 4 * g));
                            f32x4 v = acc[rt][ct];
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro_rows, (int)off, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
                        for (int ct = 0; ct < NT3; ++ct)
                            m[p * RTC + rt][ct] = fmax_raw(fmax3_raw(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2]), acc[rt][ct][3]);
                }
            }
            stamp(it, 4);
            if constexpr (!ROWS) {
                e_b = cc.b;
                e_pos0 = cc.t * TM;
                e_pending = true;
#if !SA_DEFER_EPILOGUE
                epilogue();
#endif
            }
        }
        stamp(it, 5);
        __syncthreads();  // B2: H1[next] complete, H2 reusable
        stamp(it, 6);
    }
#if SA_DEFER_EPILOGUE
    if (!ROWS) epilogue();  // the last tile's (a workgroup without tiles writes nothing: e_pending is false)
#endif
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K, int MODE>
__global__ void __launch_bounds__(512, MINW)
sa_mlp_max_kernel(const SaArgs A, const int grid) {
    sa_body<C1, C2, C3, WC, RTC, MINW, NB1, K, MODE>(A, (int)__builtin_amdgcn_workgroup_id_x(), grid);
}

// Both scales of a keypoint-query module (K0 and K1 neighbours, same layer widths) in ONE persistent grid: workgroups
// [0, n0) own the tiles of problem 0, the rest those of problem 1, each loading only its own scale's weights.  Served
// separately the K=16 scale has 384 tiles for 256 CUs (every workgroup pulls 160 KB of weights into registers to use them
// twice, 0.35 of the MFMA peak) and the K=64 scale runs 1344 tiles on 224 workgroups; together the 1728 tiles fill all
// CUs with ~7 tiles per workgroup, and a launch per module disappears.
template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K0, int K1, int MODE>
__global__ void __launch_bounds__(512, MINW)
sa_mlp_max_pair_kernel(const SaArgs A0, const SaArgs A1, const int n0, const int n1) {
    const int wg = (int)__builtin_amdgcn_workgroup_id_x();
    if (wg < n0)
        sa_body<C1, C2, C3, WC, RTC, MINW, NB1, K0, MODE>(A0, wg, n0);
    else
        sa_body<C1, C2, C3, WC, RTC, MINW, NB1, K1, MODE>(A1, wg - n0, n1);
}

// CUs a persistent SA grid may occupy.  A workgroup of this kernel fills a CU (8 waves x 256 VGPRs, ~135 KB LDS): nothing of
// a concurrent stream can share it, so a serving loop that keeps several batches in flight does better when the SA grids
// leave a few CUs to the other stream's latency-bound kernels (measured, two batches in flight: 256 CUs 0.824 ms/step,
// 240 CUs 0.815, 224 CUs 0.822; single stream 1.030 / 1.058 / 1.067).  pn2x_sa_set_compute_units(n) / PN2_SA_CUS=n; 0 = all.
static int g_sa_cus = -1;
static int sa_compute_units() {
    if (g_sa_cus < 0) {
        const char *e = getenv("PN2_SA_CUS");
        g_sa_cus = e ? atoi(e) : 0;
    }
    const int all = num_compute_units();
    return (g_sa_cus > 0 && g_sa_cus < all) ? g_sa_cus : all;
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K, int MODE>
static int launch_sa_km(int b, SaArgs a, hipStream_t st) {
    constexpr int WP = 4 / WC, TM = WP * 64;
    const int sk = a.S * a.K;
    a.tiles_per_cloud = (sk + TM - 1) / TM;
    const long num_tiles_l = (long)b * a.tiles_per_cloud;
    if (num_tiles_l > 2147483647L) return PN2_ERANGE;
    a.num_tiles = (int)num_tiles_l;
    a.B = b;
    a.lgK = 0;
    while ((1 << a.lgK) < a.K) ++a.lgK;
    const size_t lds = (size_t)TM * (NB1 * (C1 + SA_PAD) + C2 + SA_PAD) * sizeof(float);
    auto kfn = sa_mlp_max_kernel<C1, C2, C3, WC, RTC, MINW, NB1, K, MODE>;
    static PerDeviceOnce raised;  // once per instantiation and device; never during a later stream capture
    if (lds > 64 * 1024 && raised.first_use())
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // persistent workgroups: weights are loaded into registers once per workgroup
    const int wg_per_cu = (int)((160 * 1024) / lds) < MINW / 2 ? (int)((160 * 1024) / lds) : MINW / 2;
    const int max_wg = sa_compute_units() * (wg_per_cu < 1 ? 1 : wg_per_cu);
    // Balanced persistent grid: every workgroup runs the same number of tiles (ceil(tiles / rounds)), so the launch
    // takes `rounds` tile-times either way but leaves the CUs a ragged last round would idle to concurrent streams
    // (1344 tiles: 224 workgroups x 6 instead of 256 of which 64 run 6 and 192 run 5).
    const int rounds = (a.num_tiles + max_wg - 1) / max_wg;
    const int grid = (a.num_tiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, a, grid);
    return check_launch();
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1>
static void sa_tiles(int b, SaArgs &a) {
    constexpr int WP = 4 / WC, TM = WP * 64;
    a.tiles_per_cloud = (a.S * a.K + TM - 1) / TM;
    a.num_tiles = b * a.tiles_per_cloud;
    a.B = b;
    a.lgK = 0;
    while ((1 << a.lgK) < a.K) ++a.lgK;
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K0, int K1, int MODE>
static int launch_sa_pair(int b, SaArgs a0, SaArgs a1, hipStream_t st) {
    constexpr int WP = 4 / WC, TM = WP * 64;
    if ((long)b * ((a0.S * a0.K + TM - 1) / TM) + (long)b * ((a1.S * a1.K + TM - 1) / TM) > 2147483647L) return PN2_ERANGE;
    sa_tiles<C1, C2, C3, WC, RTC, MINW, NB1>(b, a0);
    sa_tiles<C1, C2, C3, WC, RTC, MINW, NB1>(b, a1);
    const size_t lds = (size_t)TM * (NB1 * (C1 + SA_PAD) + C2 + SA_PAD) * sizeof(float);
    auto kfn = sa_mlp_max_pair_kernel<C1, C2, C3, WC, RTC, MINW, NB1, K0, K1, MODE>;
    static PerDeviceOnce raised;
    if (lds > 64 * 1024 && raised.first_use())
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int wg_per_cu = (int)((160 * 1024) / lds) < MINW / 2 ? (int)((160 * 1024) / lds) : MINW / 2;
    const int max_wg = sa_compute_units() * (wg_per_cu < 1 ? 1 : wg_per_cu);
    // same number of rounds for both shares (a tile costs the same in either: 64 positions through the same layers)
    const int total = a0.num_tiles + a1.num_tiles;
    int rounds = (total + max_wg - 1) / max_wg;
    int n0, n1;
    for (;; ++rounds) {
        n0 = (a0.num_tiles + rounds - 1) / rounds;
        n1 = (a1.num_tiles + rounds - 1) / rounds;
        if (n0 + n1 <= max_wg || rounds > total) break;
    }
    hipLaunchKernelGGL(kfn, dim3(n0 + n1), dim3(512), lds, st, a0, a1, n0, n1);
    return check_launch();
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1, int K>
static int launch_sa_k(int b, const SaArgs &a, hipStream_t st) {
    const bool fa = a.a1f != nullptr, fx = a.xyz != nullptr, fc = a.cadd != nullptr;
    if (!fa && fx && !fc) return launch_sa_km<C1, C2, C3, WC, RTC, MINW, NB1, K, 0>(b, a, st);
    if (fa && fx && !fc) return launch_sa_km<C1, C2, C3, WC, RTC, MINW, NB1, K, 1>(b, a, st);
    if (fa && fx && fc) return launch_sa_km<C1, C2, C3, WC, RTC, MINW, NB1, K, 2>(b, a, st);
    return launch_sa_km<C1, C2, C3, WC, RTC, MINW, NB1, K, 3>(b, a, st);
}

template <int C1, int C2, int C3, int WC, int RTC, int MINW, int NB1>
static int launch_sa(int b, const SaArgs &a, hipStream_t st) {
    switch (a.K) {
        case 16: return launch_sa_k<C1, C2, C3, WC, RTC, MINW, NB1, 16>(b, a, st);
        case 32: return launch_sa_k<C1, C2, C3, WC, RTC, MINW, NB1, 32>(b, a, st);
        case 64: return launch_sa_k<C1, C2, C3, WC, RTC, MINW, NB1, 64>(b, a, st);
    }
    return PN2_ERANGE;
}

#ifndef SA_RTC128
#define SA_RTC128 2  /* row tiles live at once in the 128-128-192 instance; 4 = one pass per layer, but spills (measured: see profiles) */
#endif
#ifndef SA_NB1
#define SA_NB1 2  /* H1 ring depth; 3 measured no better (profiles/r01_misc_measurements.md) */
#endif
static long long *g_sa_trace = nullptr;
// The gathers address memory through 32-bit buffer offsets formed with 24-bit multiplies (sa_body): the a1f / xyz tensors
// (one descriptor each, scalar cloud offset) and every cloud's idx / cxyz / cadd block must stay below 4 GiB, row indices
// and row strides (bytes) below 2^24.
static bool sa_ranges_ok(long b, long n, long s, long k, long a1f_ld, long cadd_ld, long c3, long out_s, long out_c) {
    const long lim = 0xffffffffL;
    if (out_s < 0 || out_c < 0 || 4 * ((s - 1) * out_s + (c3 - 1) * out_c + 1) >= lim) return false;  // one store descriptor per cloud
    return n < (1L << 24) && s < (1L << 24) && 4 * a1f_ld < (1L << 24) && 4 * cadd_ld < (1L << 24) &&
           4 * b * n * (a1f_ld > 3 ? a1f_ld : 3) <= lim && 4 * s * (cadd_ld > 3 ? cadd_ld : 3) <= lim && 4 * s * k <= lim;
}
}  // namespace pn2

extern "C" int pn2x_sa_set_compute_units(int n) {
    if (n < 0) return PN2_EINVAL;
    pn2::g_sa_cus = n;
    return PN2_OK;
}

extern "C" void pn2x_debug_set_sa_trace(void *device_buffer_2x8x8_int64) { pn2::g_sa_trace = (long long *)device_buffer_2x8x8_int64; }

extern "C" int pn2x_sa_mlp_max(int b, int n, int s, int k, int c1, int c2, int c3, const float *a1f, int a1f_ld,
                               const float *xyz, const float *cxyz, const float *wx, const float *b1, const float *cadd,
                               int cadd_ld, const int *idx,
                               const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                               long out_b, int out_s, int out_c, void *stream) {
    using namespace pn2;
    if (b < 0 || n < 1 || s < 0 || k < 1) return PN2_EINVAL;
    if (b == 0 || s == 0) return PN2_OK;
    if (!idx || !w2 || !b2 || !w3 || !b3 || !out) return PN2_ENULL;
    if (!a1f && !xyz) return PN2_ENULL;          // layer 1 needs at least one per-point term
    if (xyz && (!cxyz || !wx)) return PN2_ENULL;
    if ((a1f && (a1f_ld < c1 || a1f_ld % 4)) || (cadd && (cadd_ld < c1 || cadd_ld % 4))) return PN2_EINVAL;
    if (!(k == 16 || k == 32 || k == 64)) return PN2_ERANGE;
    if (!sa_ranges_ok(b, n, s, k, a1f ? a1f_ld : 0, cadd ? cadd_ld : 0, c3, out_s, out_c)) return PN2_ERANGE;
    if (((uintptr_t)a1f | (uintptr_t)cadd | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)w3) % 16 != 0) return PN2_EINVAL;
    SaArgs a;
    a.B = b; a.N = n; a.S = s; a.K = k; a.lgK = 0;
    a.a1f = a1f; a.a1f_ld = a1f_ld; a.cadd_ld = cadd_ld; a.xyz = xyz; a.cxyz = cxyz; a.wx = wx; a.b1 = b1; a.cadd = cadd; a.idx = idx;
    a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.w2e = nullptr; a.out = out; a.out_b = out_b; a.out_s = out_s; a.out_c = out_c;
    a.num_tiles = 0; a.tiles_per_cloud = 0; a.trace = g_sa_trace;
    hipStream_t st = (hipStream_t)stream;
    if (c1 == 32 && c2 == 32 && c3 == 64) return launch_sa<32, 32, 64, 2, 4, 4, SA_NB1>(b, a, st);
    if (c1 == 64 && c2 == 64 && c3 == 128) return launch_sa<64, 64, 128, 4, 2, 4, SA_NB1>(b, a, st);
    if (c1 == 128 && c2 == 128 && c3 == 192) return launch_sa<128, 128, 192, 4, SA_RTC128, 2, SA_NB1>(b, a, st);
    return PN2_ERANGE;
}

namespace pn2 {
static int fill_sa_args(int b, const pn2x_sa_problem &p, int c1, int c3, SaArgs &a) {
    if (p.n < 1 || p.s < 1 || p.k < 1) return PN2_EINVAL;
    if (!p.idx || !p.w2 || !p.b2 || !p.w3 || !p.b3 || !p.out) return PN2_ENULL;
    if (!p.a1f && !p.xyz) return PN2_ENULL;
    if (p.xyz && (!p.cxyz || !p.wx)) return PN2_ENULL;
    if ((p.a1f && (p.a1f_ld < c1 || p.a1f_ld % 4)) || (p.cadd && (p.cadd_ld < c1 || p.cadd_ld % 4))) return PN2_EINVAL;
    if (((uintptr_t)p.a1f | (uintptr_t)p.cadd | (uintptr_t)p.b1 | (uintptr_t)p.w2 | (uintptr_t)p.w3) % 16 != 0) return PN2_EINVAL;
    if (!sa_ranges_ok(b, p.n, p.s, p.k, p.a1f ? p.a1f_ld : 0, p.cadd ? p.cadd_ld : 0, c3, p.out_s, p.out_c)) return PN2_ERANGE;
    a.B = 0; a.N = p.n; a.S = p.s; a.K = p.k; a.lgK = 0;
    a.a1f = p.a1f; a.a1f_ld = p.a1f_ld; a.cadd_ld = p.cadd_ld; a.xyz = p.xyz; a.cxyz = p.cxyz; a.wx = p.wx; a.b1 = p.b1;
    a.cadd = p.cadd; a.idx = p.idx; a.w2 = p.w2; a.b2 = p.b2; a.w3 = p.w3; a.b3 = p.b3; a.w2e = nullptr; a.out = p.out; a.out_b = p.out_b;
    a.out_s = p.out_s; a.out_c = p.out_c; a.num_tiles = 0; a.tiles_per_cloud = 0; a.trace = nullptr;
    return PN2_OK;
}
}  // namespace pn2

extern "C" int pn2x_sa_mlp_max_pair_supported(int k0, int k1, int c1, int c2, int c3) {
    return (c1 == 128 && c2 == 128 && c3 == 192 && ((k0 == 16 && k1 == 64) || (k0 == 64 && k1 == 16))) ? 1 : 0;
}

extern "C" int pn2x_sa_mlp_max_pair(int b, int c1, int c2, int c3, const pn2x_sa_problem *p0, const pn2x_sa_problem *p1, void *stream) {
    using namespace pn2;
    if (b < 0 || !p0 || !p1) return b < 0 ? PN2_EINVAL : PN2_ENULL;
    if (b == 0) return PN2_OK;
    if (!pn2x_sa_mlp_max_pair_supported(p0->k, p1->k, c1, c2, c3)) return PN2_ERANGE;
    if (p0->k > p1->k) { const pn2x_sa_problem *t = p0; p0 = p1; p1 = t; }
    SaArgs a0, a1;
    int rc = fill_sa_args(b, *p0, c1, c3, a0);
    if (rc != PN2_OK) return rc;
    rc = fill_sa_args(b, *p1, c1, c3, a1);
    if (rc != PN2_OK) return rc;
    const bool fa = a0.a1f && a1.a1f, fx = a0.xyz && a1.xyz, fc0 = a0.cadd != nullptr, fc1 = a1.cadd != nullptr;
    if (!fa || !fx || fc0 != fc1 || (a0.a1f == nullptr) != (a1.a1f == nullptr)) return PN2_ERANGE;  // both scales: a1f + xyz (+ cadd)
    hipStream_t st = (hipStream_t)stream;
    if (fc0) return launch_sa_pair<128, 128, 192, 4, SA_RTC128, 2, SA_NB1, 16, 64, 2>(b, a0, a1, st);
    return launch_sa_pair<128, 128, 192, 4, SA_RTC128, 2, SA_NB1, 16, 64, 1>(b, a0, a1, st);
}

// ---- two-layer MLP over plain rows: out[r, :] = relu(W3 relu(W2 x[r, :] + b2) + b3) -----------------------------------------
// The tile loop of the set-abstraction kernel without neighbourhoods (MODE 4): the intermediate activation never leaves the
// chip.  Used for consecutive per-point layers whose two weight matrices fit the register file (feature propagation).
extern "C" int pn2x_mlp2_rows_supported(int c1, int c2, int c3) { return (c1 == 128 && c2 == 128 && c3 == 128) ? 1 : 0; }

extern "C" int pn2x_mlp2_rows(long rows, int c1, int c2, int c3, const float *x, int ldx, const float *w2, const float *w2e, const float *b2,
                              const float *w3, const float *b3, float *out, int ldo, void *stream) {
    using namespace pn2;
    if (rows < 0 || ldx < c1 + (w2e ? 4 : 0) || ldo < c3 || ldx % 4 || ldo % 4) return PN2_EINVAL;  // (16-byte row segments in and out)
    if (rows == 0) return PN2_OK;
    if (!x || !w2 || !b2 || !w3 || !b3 || !out) return PN2_ENULL;
    if (!pn2x_mlp2_rows_supported(c1, c2, c3)) return PN2_ERANGE;
    if (((uintptr_t)x | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)out) % 16 != 0) return PN2_EINVAL;
    if (rows >= (1L << 24) - 64 || 4L * ldx >= (1L << 24) || 4L * rows * ldx > 0xffffffffL || 4L * rows * ldo > 0xffffffffL) return PN2_ERANGE;
    SaArgs a;
    a.B = 1; a.N = (int)rows; a.K = 16; a.S = (int)((rows + 15) / 16); a.lgK = 4;
    a.a1f = x; a.a1f_ld = ldx; a.cadd_ld = 0; a.xyz = nullptr; a.cxyz = nullptr; a.wx = nullptr; a.b1 = nullptr; a.cadd = nullptr; a.idx = nullptr;
    a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.w2e = w2e; a.out = out; a.out_b = 0; a.out_s = ldo; a.out_c = 1;
    a.num_tiles = 0; a.tiles_per_cloud = 0; a.trace = nullptr;
    if (w2e) return launch_sa_km<128, 128, 128, 4, 2, 2, SA_NB1, 16, 5>(1, a, (hipStream_t)stream);
    return launch_sa_km<128, 128, 128, 4, 2, 2, SA_NB1, 16, 4>(1, a, (hipStream_t)stream);
}

extern "C" int pn2x_sa_mlp_max_supported(int k, int c1, int c2, int c3) {
    const bool kk = (k == 16 || k == 32 || k == 64);
    const bool cc = (c1 == 32 && c2 == 32 && c3 == 64) || (c1 == 64 && c2 == 64 && c3 == 128) ||
                    (c1 == 128 && c2 == 128 && c3 == 192);
    return (kk && cc) ? 1 : 0;
}
