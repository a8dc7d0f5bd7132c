/* sdf_oracle.c -- CPU restatement of the particle optimisers' SDF lookups.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pn2_oracle.c): may be used by tests/, __graft_entry__.smoke()
 * and bench-side cpu_baseline legs; never by anything under hotrack_amd/ or network/.
 *
 * Parity: PINNED against the reference itself.  The reference's Python for this path imports and runs on
 * CPU in the build container (tests/golden/make_golden_sdf.py, harness-side stubs for cv2/open3d/...);
 * its outputs on seeded inputs are committed under tests/golden/sdf_*.npz and this file is checked against
 * them by tests/test_sdf_oracle.py:
 *   - pn2o_sdf_trilinear      == gf_optimize_obj.Distance          bit-for-bit
 *   - pn2o_sdf_nearest        == gf_optimize_hand_pose.query_sdf   bit-for-bit on the voxel, except where the
 *                                reference's 3x3 matmul (BLAS, unspecified accumulation order) lands a
 *                                coordinate on the other side of a voxel face (measured, bounded in the test)
 *   - pn2o_sdf_particle_energy == gf_optimize_obj.evaluate         <= 1e-6 (matmul / mean summation order)
 *
 * Arithmetic: every elementwise torch op of the reference is one correctly rounded fp32 operation; this
 * file performs the same operations in the same order (compiled with -ffp-contract=off).  The only fused
 * operations are the explicit fmaf() chains of the 3x3 transforms, which the HIP kernels repeat exactly.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PN2O_OK 0
#define PN2O_EINVAL (-1)

/* IEEE binary16 -> binary32 (exact). */
static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline float vol_at(const void *vol, int is_f16, long long i) {
    return is_f16 ? half_to_float(((const uint16_t *)vol)[i]) : ((const float *)vol)[i];
}

static inline float clampf(float v, float lo, float hi) { /* torch.clamp: min(max(v, lo), hi) */
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

static inline long long clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One point of gf_optimize_obj.Distance (optimization_obj.py:184-228). */
static float trilinear_one(float vx, float vy, float vz, const void *vol, int is_f16, int res, float bbox_min,
                           float stride, float clamp_lo, float clamp_hi) {
    const float top = (float)(res - 1);
    float x = clampf((vx - bbox_min) / stride, 0.0f, top); /* :191-196 */
    float y = clampf((vy - bbox_min) / stride, 0.0f, top);
    float z = clampf((vz - bbox_min) / stride, 0.0f, top);
    long long xi = (long long)x, yi = (long long)y, zi = (long long)z; /* :199-201, x >= 0 -> trunc == floor */
    x -= (float)xi; /* :203-205 */
    y -= (float)yi;
    z -= (float)zi;
    const long long R = res, last = R * R * R - 1;
    long long i000 = (xi * R + yi) * R + zi; /* :206-213 */
    long long i001 = i000 + 1, i010 = i000 + R, i011 = i001 + R;
    long long i100 = i000 + R * R, i101 = i001 + R * R, i110 = i010 + R * R, i111 = i011 + R * R;
    float d000 = vol_at(vol, is_f16, clampll(i000, 0, last)), d001 = vol_at(vol, is_f16, clampll(i001, 0, last));
    float d010 = vol_at(vol, is_f16, clampll(i010, 0, last)), d011 = vol_at(vol, is_f16, clampll(i011, 0, last));
    float d100 = vol_at(vol, is_f16, clampll(i100, 0, last)), d101 = vol_at(vol, is_f16, clampll(i101, 0, last));
    float d110 = vol_at(vol, is_f16, clampll(i110, 0, last)), d111 = vol_at(vol, is_f16, clampll(i111, 0, last));
    const float mx = 1.0f - x, my = 1.0f - y, mz = 1.0f - z;
    /* :223-226, same association as the reference expression */
    float lo = ((d000 * mz + d001 * z) * my + (d010 * mz + d011 * z) * y) * mx;
    float hi = ((d100 * mz + d101 * z) * my + (d110 * mz + d111 * z) * y) * x;
    return clampf(lo + hi, clamp_lo, clamp_hi); /* :227 */
}

/* V (m,3) -> out (m).  vol: res^3 values, fp16 bits (is_f16) or fp32. */
int pn2o_sdf_trilinear(int m, const float *V, const void *vol, int is_f16, int res, float bbox_min, float stride,
                       float clamp_lo, float clamp_hi, float *out) {
    if (m < 0 || res < 2 || !(stride > 0.0f)) return PN2O_EINVAL;
    for (int i = 0; i < m; ++i)
        out[i] = trilinear_one(V[3 * i], V[3 * i + 1], V[3 * i + 2], vol, is_f16, res, bbox_min, stride, clamp_lo, clamp_hi);
    return PN2O_OK;
}

/* (p - t) @ R with the fixed chain  o_j = fma(q2, R[2][j], fma(q1, R[1][j], q0 * R[0][j])). */
static inline void to_object_frame(const float *p, const float *t, const float *R, float *o) {
    const float q0 = p[0] - t[0], q1 = p[1] - t[1], q2 = p[2] - t[2];
    for (int j = 0; j < 3; ++j) o[j] = fmaf(q2, R[6 + j], fmaf(q1, R[3 + j], q0 * R[j]));
}

/* gf_optimize_obj.evaluate (optimization_obj.py:230-237): pcld (n,3) shared by all particles, rot (P,3,3),
 * trans (P,3) -> sdf_energy (P) = mean_n |Distance((pcld - t_p) @ R_p)|.  Mean accumulated in double. */
int pn2o_sdf_particle_energy(int p, int n, const float *pcld, const float *rot, const float *trans, const void *vol,
                             int is_f16, int res, float bbox_min, float stride, float clamp_lo, float clamp_hi,
                             float *sdf_energy) {
    if (p < 0 || n < 1 || res < 2 || !(stride > 0.0f)) return PN2O_EINVAL;
    for (int i = 0; i < p; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) {
            float o[3];
            to_object_frame(pcld + 3 * j, trans + 3 * i, rot + 9 * i, o);
            acc += (double)fabsf(trilinear_one(o[0], o[1], o[2], vol, is_f16, res, bbox_min, stride, clamp_lo, clamp_hi));
        }
        sdf_energy[i] = (float)(acc / (double)n);
    }
    return PN2O_OK;
}

/* c10::div_floor_floating (PyTorch c10/util/generic_math.h) -- what `tensor // scalar` computes for floats. */
static float div_floor_f32(float a, float b) {
    if (b == 0.0f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    float floordiv;
    if (div != 0.0f) {
        floordiv = floorf(div);
        if (div - floordiv > 0.5f) floordiv += 1.0f;
    } else {
        floordiv = copysignf(0.0f, a / b);
    }
    return floordiv;
}

int pn2o_div_floor(int m, const float *a, float b, float *out) { /* exposed so tests can pin it to torch's `//` */
    for (int i = 0; i < m; ++i) out[i] = div_floor_f32(a[i], b);
    return PN2O_OK;
}

/* gf_optimize_hand_pose.query_sdf (optimization_hand.py:252-262) (+ get_penetration_loss :264-268).
 * hand (b,n,3); obj_r (3,3); obj_t (3); vol res^3 (odd res), element i = (ix*res + iy)*res + iz.
 * out_idx (b,n) int32 flat voxel index or NULL; out_sdf (b,n) same element type as vol, or NULL;
 * out_pen (b) = max_n |sdf| * (sdf < 0), element type of vol, or NULL. */
int pn2o_sdf_nearest(int b, int n, const float *hand, const float *obj_r, const float *obj_t, const void *vol,
                     int is_f16, int res, float voxel_scale, int *out_idx, void *out_sdf, void *out_pen) {
    if (b < 0 || n < 1 || res < 1 || !(voxel_scale > 0.0f)) return PN2O_EINVAL;
    const int half = res / 2;
    for (int i = 0; i < b; ++i) {
        float pen = 0.0f;
        uint16_t pen_bits = 0;
        for (int j = 0; j < n; ++j) {
            float o[3];
            long long id[3];
            to_object_frame(hand + ((size_t)i * n + j) * 3, obj_t, obj_r, o);
            for (int a = 0; a < 3; ++a)
                id[a] = (long long)clampf(div_floor_f32(o[a], voxel_scale), (float)-half, (float)half) + half; /* :255-257 */
            if (id[0] >= res || id[1] >= res || id[2] >= res) return PN2O_EINVAL; /* the asserts at :258-260 (even res) */
            const long long flat = (id[0] * res + id[1]) * res + id[2];
            if (out_idx) out_idx[(size_t)i * n + j] = (int)flat;
            if (is_f16) {
                const uint16_t h = ((const uint16_t *)vol)[flat];
                if (out_sdf) ((uint16_t *)out_sdf)[(size_t)i * n + j] = h;
                const float v = half_to_float(h);
                if (v < 0.0f && -v > pen) { /* |v| * (v < 0), max over n (:265-267); volumes are NaN-free */
                    pen = -v;
                    pen_bits = (uint16_t)(h & 0x7fffu);
                }
            } else {
                const float v = ((const float *)vol)[flat];
                if (out_sdf) ((float *)out_sdf)[(size_t)i * n + j] = v;
                if (v < 0.0f && -v > pen) pen = -v;
            }
        }
        if (out_pen) {
            if (is_f16)
                ((uint16_t *)out_pen)[i] = pen_bits;
            else
                ((float *)out_pen)[i] = pen;
        }
    }
    return PN2O_OK;
}
