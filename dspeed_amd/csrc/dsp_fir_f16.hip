// dsp_fir_f16.hip -- the long FIR of the energy filters (convolve_wf 'v' + numpy.amax: dsp_fir_mfma.hip's product) on the HALF-PRECISION
// matrix instructions, float32-accurate.
//
// v_mfma_f32_16x16x4_f32 is an exact float32 chain but runs at 1/16 of the rate of v_mfma_f32_16x16x32_f16.  A float32 number is the sum
// of two float16 numbers to 22 bits -- hi = half(x), lo = half(x - hi) -- once it is scaled into float16's range by a power of two, so
//     x . t  =  hi_x hi_t  +  hi_x lo_t  +  lo_x hi_t  (+ lo_x lo_t, 2^-22 of the product: dropped)
// is three float16 products, each exact in the instruction's float32 accumulator: 3/16 of the matrix time of the float32 form, and the
// representation error (measured on the cusp / zac kernels and BASELINE's rows: 4e-9 .. 2e-7 of the filtered waveform's peak,
// tools/fir_accuracy.py) is below what accumulating in float32 costs either way.  As in the float32 kernel the partial sums leave float32
// every 256 samples and are added in float64.
//
//   * rows: every row gets its own power-of-two scale from its largest |sample - baseline| (a first pass of the workgroup over its 64
//     rows), so the split keeps 22 bits whatever the waveform's magnitude; the scale is undone, exactly, on the way out;
//   * taps: dsp_fir_f16_prep_kernel scales the kernel by one power of two, splits it, reverses it and writes it with zero margins EIGHT
//     times, shifted by 0..7 elements: the B fragment of lane (column c, k-block h) is the 8 consecutive taps kr[k + 8h - c ..], a 16-byte
//     read that is aligned in the copy shifted by (-c) mod 8 -- the Toeplitz matrix is never materialised;
//   * per stage of 64 samples the 64 x 64 A tile (two float16 planes) and the 384-tap window of the 16 tap copies are staged in LDS,
//     double buffered; a wavefront owns 32 rows x 80 columns = 2 x 5 tiles and issues 30 MFMAs per 32 samples against 14 ds_read_b128;
//   * rows holding a NaN / an infinity / a sample beyond float16's scaled range are handled as in the float32 kernel (NaN, or tap by tap).
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

#define FIR_LDS __attribute__((address_space(3)))
#define FIR_GLOBAL __attribute__((address_space(1)))
#define FIR_KARG __attribute__((address_space(4)))

namespace {

constexpr int BM = 64, BN = 320, BK = 64, MT = 2, NT = 5;
constexpr int APITCH = BK + 8;        // halfs per A row: 144 bytes, rows 16 bytes apart modulo the 128-byte bank line
constexpr int TWIN = BN + BK;         // taps a stage's fragments can reach
constexpr int TPITCH = TWIN + 8;      // halfs per tap copy in LDS: copies 16 bytes apart modulo the bank line
constexpr int KFLUSH = 256;           // samples between two float64 flushes

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// ---- taps: scaled, split, reversed, zero-margined, eight shifted copies.  Layout of `dst` (halfs): [split 0..1][shift 0..7][TZ], then one float
// (the inverse scale) at byte offset 2 * 16 * TZ.  tapz[idx] = kr[idx - BN] for 0 <= idx - BN < m, else 0;  copy_r[i] = tapz[i + r].
__global__ void __launch_bounds__(256) dsp_fir_f16_prep_kernel(const float* __restrict__ taps, int m, int TZ, _Float16* __restrict__ dst) {
    __shared__ float red[256];
    const int tid = (int)threadIdx.x;
    float mx = 0.0f;
    for (int t = tid; t < m; t += 256) {
        const float a = __builtin_fabsf(taps[t]);
        mx = a > mx ? a : mx;
    }
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;  // floor(log2(mx)) for normal numbers
    if (!(mx > 0.0f) || e < -100) e = 14;                        // all zeros (or denormal noise): scale 1
    const float scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);      // mx * scale in [2^14, 2^15)
    const float inv_scale = __uint_as_float((unsigned)(127 - 14 + e) << 23);
    for (int idx = tid; idx < 8 * TZ; idx += 256) {
        const int r = idx / TZ, i = idx - r * TZ;
        const int t = i + r - BN;
        const float v = (t >= 0 && t < m) ? taps[m - 1 - t] * scale : 0.0f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        dst[(size_t)r * TZ + i] = hi;
        dst[(size_t)(8 + r) * TZ + i] = lo;
    }
    if (tid == 0) *(float*)(dst + (size_t)16 * TZ) = inv_scale;
}

template <int IN>
__global__ void __launch_bounds__(512, 1) dsp_fir_f16_kernel(FirArgs A_, FirF16Taps T_, int64_t n_wf) {
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) unsigned char f16_smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 4 wavefronts: rows 32 wm .., columns 80 wn ..
    const int q = (int)blockIdx.x;            // which kernel
    const int64_t row0 = (int64_t)blockIdx.y * BM;
    const int n = A.n, m = A.m[q], p = A.p[q], kend = A.kend;
    const int TZ = T_.tz;
    const FIR_GLOBAL _Float16* tg = (const FIR_GLOBAL _Float16*)T_.taps16[q];
    const float tap_inv = *(const FIR_GLOBAL float*)(tg + (size_t)16 * TZ);

    // LDS: A planes [buf][split][BM][APITCH], tap windows [buf][split * 8 + shift][TPITCH], reduction scratch
    FIR_LDS _Float16* As = (FIR_LDS _Float16*)f16_smem;
    FIR_LDS _Float16* Tw = As + 2 * 2 * BM * APITCH;
    FIR_LDS float* red = (FIR_LDS float*)(Tw + 2 * 16 * TPITCH);  // [BM][4][2]
    FIR_LDS float* rscale = red + BM * 4 * 2;                      // [BM]: 1 / the row's scale

    // ---- staging geometry: thread t carries 8 consecutive samples of row t / 8
    const int srow = tid >> 3, skc = (tid & 7) * 8;
    const int64_t grow = row0 + srow < n_wf ? row0 + srow : n_wf - 1;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (grow * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[grow * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    bool bad = false, has_nan = false;

    auto load8 = [&](int k, float (&x)[8]) {  // 8 samples from sample k of the thread's row, baseline subtracted, zeros beyond the slice
        if (IN == 0) {
            const f4 v0 = *(const FIR_GLOBAL f4*)(rowp + (size_t)k * 4), v1 = *(const FIR_GLOBAL f4*)(rowp + (size_t)k * 4 + 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = v0[u];
                x[4 + u] = v1[u];
            }
        } else {
            const u4 raw = *(const FIR_GLOBAL u4*)(rowp + (size_t)k * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[2 * u] = IN == 1 ? (float)(short)(raw[u] & 0xffffu) : (float)(raw[u] & 0xffffu);
                x[2 * u + 1] = IN == 1 ? (float)(short)(raw[u] >> 16) : (float)(raw[u] >> 16);
            }
        }
        const int live = n - k;  // samples of this vector inside the slice (>= 8 almost always)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float v = sub ? x[u] - bl : x[u];
            if (live < 8 && u >= live) v = 0.0f;
            x[u] = v;
        }
    };

    // ---- pass 0: the row's scale.  The 8 threads of a row walk it 64 samples apart and agree on the largest magnitude.
    float mx = 0.0f;
    for (int k = skc; k < kend; k += 4 * BK) {  // (four loads in flight: the pass is a latency chain otherwise)
        float x[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (k + b * BK < kend) load8(k + b * BK, x[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (k + b * BK < kend) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a = __builtin_fabsf(x[b][u]);
                    has_nan |= (x[b][u] != x[b][u]);
                    mx = a > mx ? a : mx;  // (a NaN never raises it)
                }
            }
    }
#pragma unroll
    for (int sft = 1; sft < 8; sft <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    // an infinity, or a magnitude so near float32's limits that the scale itself would leave them: the row goes the slow way (scale 1 here)
    bad = !(mx <= 3.4028234663852886e38f) || (mx > 0.0f && (e < -100 || e > 100));
    if (!(mx > 0.0f) || bad) e = 14;
    const float xs = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    if ((tid & 7) == 0) rscale[srow] = __uint_as_float((unsigned)(127 - 14 + e) << 23) * tap_inv;

    float stage_x[8];
    bool stage_live = false;
    auto fetch = [&](int k0) {
        const int k = k0 + skc;
        stage_live = k < kend;  // (kend is a multiple of 32, the thread's 8 samples lie on one side of it)
        if (stage_live) load8(k, stage_x);
    };
    auto commit = [&](int buf) {  // scaled, split, into the two planes
        h8 hi, lo;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float v = stage_live ? stage_x[u] * xs : 0.0f;
            const _Float16 h = (_Float16)v;
            hi[u] = h;
            lo[u] = (_Float16)(v - (float)h);
        }
        FIR_LDS _Float16* ap = As + (size_t)buf * 2 * BM * APITCH + srow * APITCH + skc;
        *(FIR_LDS h8*)ap = hi;
        *(FIR_LDS h8*)(ap + BM * APITCH) = lo;
    };
    // the 16 tap copies' windows [k0, k0 + TWIN): 16 x 48 vectors of 8 halfs, 768 vectors over 512 threads
    h8 tap_v[2];
    int tap_src[2], tap_dst[2];  // (this thread's two vectors: where they come from in the image, where they go in the window)
    bool tap_live[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = tid + it * 512;
        const int c = v / (TWIN / 8), o = (v - c * (TWIN / 8)) * 8;
        tap_live[it] = v < 16 * (TWIN / 8);
        tap_src[it] = c * TZ + o;
        tap_dst[it] = c * TPITCH + o;
    }
    auto fetch_taps = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (tap_live[it]) tap_v[it] = *(const FIR_GLOBAL h8*)(tg + tap_src[it] + k0);
    };
    auto commit_taps = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (tap_live[it]) *(FIR_LDS h8*)(Tw + buf * 16 * TPITCH + tap_dst[it]) = tap_v[it];
    };

    typedef float acc_t __attribute__((ext_vector_type(4)));
    acc_t acc[MT][NT];
    double tot[MT][NT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            acc[a][b] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[a][b][r] = 0.0;
        }

    const int j = lane & 15, h4 = lane >> 4;
    // A fragment: row 32 wm + 16 tm + j, samples 32 g + 8 h4 .. + 7 of the stage
    const int a_off = (wm * 32 + j) * APITCH + 8 * h4;
    // B fragment: tapz[BN + k0 + 32 g + 8 h4 - c ..], c = 80 wn + 16 tn + j: aligned in the copy shifted by r = (-j) mod 8, at window offset
    // BN + 32 g + 8 h4 - c - r
    const int shift = (8 - (j & 7)) & 7;
    const int t_off = shift * TPITCH + BN + 8 * h4 - (wn * 80 + j) - shift;

    fetch(0);
    fetch_taps(0);
    commit(0);
    commit_taps(0);
    __syncthreads();
    const int n_stage = (kend + BK - 1) / BK;
    // (two loops: the float64 flush sits between runs of KFLUSH / BK stages, outside the stage loop -- inside it, as a conditional, the
    // compiler copies all 120 accumulator registers around the branch in every stage)
    for (int st0 = 0; st0 < n_stage; st0 += KFLUSH / BK) {
        const int st1 = st0 + KFLUSH / BK < n_stage ? st0 + KFLUSH / BK : n_stage;
        for (int st = st0; st < st1; ++st) {
            const int buf = st & 1, k0 = st * BK;
#ifndef F16_NO_FETCH
            if (st + 1 < n_stage) {
                fetch(k0 + BK);
                fetch_taps(k0 + BK);
            }
#endif
            const FIR_LDS _Float16* ab = As + buf * 2 * BM * APITCH + a_off;
            const FIR_LDS _Float16* tb = Tw + buf * 16 * TPITCH + t_off;
#pragma unroll
            for (int g = 0; g < BK / 32; ++g) {
                h8 ah[MT], al[MT], bh[NT], blo[NT];
#pragma unroll
                for (int tm = 0; tm < MT; ++tm) {
                    ah[tm] = *(const FIR_LDS h8*)(ab + tm * 16 * APITCH + g * 32);
                    al[tm] = *(const FIR_LDS h8*)(ab + BM * APITCH + tm * 16 * APITCH + g * 32);
                }
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    bh[tn] = *(const FIR_LDS h8*)(tb + g * 32 - tn * 16);
                    blo[tn] = *(const FIR_LDS h8*)(tb + 8 * TPITCH + g * 32 - tn * 16);
                }
#pragma unroll
                for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], blo[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }
#ifndef F16_NO_COMMIT
            if (st + 1 < n_stage) {
                commit(buf ^ 1);
                commit_taps(buf ^ 1);
            }
#endif
            __syncthreads();
        }
        // partial sums of (up to) 256 samples leave float32 here
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tot[tm][tn][r] += (double)acc[tm][tn][r];
                acc[tm][tn] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    // ---- numpy.amax over the valid columns of every row; C layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = wm * 32 + tm * 16 + h4 * 4 + r;
            const double back = (double)rscale[rl];  // (a power of two: exact)
            float vmax = -__builtin_inff();
            bool vnan = false;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const float v = (float)((tot[tm][tn][r] + (double)acc[tm][tn][r]) * back);
                const bool valid = wn * 80 + tn * 16 + j < p;
                vnan |= valid && (v != v);
                vmax = (valid && v > vmax) ? v : vmax;
            }
            float fn = vnan ? 1.0f : 0.0f;
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                vmax = fmaxf(vmax, __shfl_xor(vmax, sft));
                fn = fmaxf(fn, __shfl_xor(fn, sft));
            }
            if (j == 0) {
                red[(rl * 4 + wn) * 2] = vmax;
                red[(rl * 4 + wn) * 2 + 1] = fn;
            }
        }
    // samples of the waveform outside the slice: bl_subtract's "NaN anywhere" covers them (DSP_OP_LOAD ip[0..1])
    if (IN == 0 && sub) {
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
            const int cnt = part == 0 ? A.scan_before : A.scan_after;
            const FIR_GLOBAL float* sp = (const FIR_GLOBAL float*)rowp + (part == 0 ? -cnt : n);
            for (int ee = tid & 7; ee < cnt; ee += 8) {
                const float x = sp[ee];
                has_nan |= (x != x);
            }
        }
    }
    // row screening: the 8 threads of a row agree
    unsigned flags = (bad ? 1u : 0u) | (has_nan ? 2u : 0u);
#pragma unroll
    for (int sft = 1; sft < 8; sft <<= 1) flags |= (unsigned)__shfl_xor((int)flags, sft);
    __syncthreads();
    FIR_LDS unsigned* rowflag = (FIR_LDS unsigned*)(As);  // (the A buffers are free now)
    if ((tid & 7) == 0) rowflag[srow] = flags;
    __syncthreads();
    FIR_GLOBAL float* outp = (FIR_GLOBAL float*)A.out[q];
    if (tid < BM && row0 + tid < n_wf) {
        float vmax = -__builtin_inff(), fn = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            vmax = fmaxf(vmax, red[(tid * 4 + w) * 2]);
            fn = fmaxf(fn, red[(tid * 4 + w) * 2 + 1]);
        }
        const unsigned f = rowflag[tid];
        float res = (fn != 0.0f || (f & 2u)) ? quiet_nan<float>() : vmax;
        if ((f & 1u) && !(f & 2u)) res = 0.0f;  // an infinity in the row: decided below
        outp[(row0 + tid) * A.out_stride[q]] = res;
    }
    // ---- rows with an infinity (no NaN): tap by tap in float32, the way np.convolve sees them
    const FIR_GLOBAL float* kp = (const FIR_GLOBAL float*)A.taps[q];
    for (int rl = 0; rl < BM; ++rl) {
        const unsigned f = rowflag[rl];  // (uniform)
        if (!(f & 1u) || (f & 2u) || row0 + rl >= n_wf) continue;
        __syncthreads();
        const FIR_GLOBAL char* rp = (const FIR_GLOBAL char*)A.wf + ((row0 + rl) * A.wf_stride + A.wf_offset) * ESZ;
        const float rbl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[(row0 + rl) * A.bl_stride] : A.bl_const) : 0.0f;
        float vmax = -__builtin_inff();
        bool vnan = false;
        for (int jo = tid; jo < p; jo += 512) {
            float s = 0.0f;
            for (int t = 0; t < m; ++t) {
                float x = IN == 0 ? ((const FIR_GLOBAL float*)rp)[jo + t]
                                  : (IN == 1 ? (float)((const FIR_GLOBAL short*)rp)[jo + t] : (float)((const FIR_GLOBAL unsigned short*)rp)[jo + t]);
                if (sub) x = x - rbl;
                s = __builtin_fmaf(x, kp[m - 1 - t], s);
            }
            vnan |= (s != s);
            vmax = s > vmax ? s : vmax;
        }
        float fn = vnan ? 1.0f : 0.0f;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            vmax = fmaxf(vmax, __shfl_xor(vmax, sft));
            fn = fmaxf(fn, __shfl_xor(fn, sft));
        }
        if (lane == 0) {
            red[wave * 2] = vmax;
            red[wave * 2 + 1] = fn;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 8; ++w) {
                vmax = fmaxf(vmax, red[w * 2]);
                fn = fmaxf(fn, red[w * 2 + 1]);
            }
            outp[(row0 + rl) * A.out_stride[q]] = fn != 0.0f ? quiet_nan<float>() : vmax;
        }
    }
}

}  // namespace

// halfs of the tap image of one kernel: 16 copies of TZ = BN + kend rounded up to a stage + the margin the shifted copies and the last
// window reach into, then the inverse scale (one float, kept 16-byte aligned)
extern "C" int dsp_internal_fir_f16_tz(int kend) { return ((kend + BK - 1) / BK) * BK + TWIN + 16; }
extern "C" size_t dsp_internal_fir_f16_taps_bytes(int kend) { return (size_t)16 * dsp_internal_fir_f16_tz(kend) * 2 + 16; }
extern "C" int dsp_internal_fir_f16_lds_bytes() { return (2 * 2 * BM * APITCH + 2 * 16 * TPITCH) * 2 + (BM * 4 * 2 + BM) * 4; }

extern "C" int dsp_internal_launch_fir_f16(const FirArgs* A, const FirF16Taps* T, int64_t n_wf, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0 || A->n_kernels <= 0) return 0;
    for (int q = 0; q < A->n_kernels; ++q)
        hipLaunchKernelGGL(dsp_fir_f16_prep_kernel, dim3(1), dim3(256), 0, stream, A->taps[q], A->m[q], T->tz, (_Float16*)T->taps16[q]);
    const dim3 grid((unsigned)A->n_kernels, (unsigned)((n_wf + BM - 1) / BM));
    switch (A->in_kind) {
        case 0: hipLaunchKernelGGL(dsp_fir_f16_kernel<0>, grid, dim3(512), lds_bytes, stream, *A, *T, n_wf); break;
        case 1: hipLaunchKernelGGL(dsp_fir_f16_kernel<1>, grid, dim3(512), lds_bytes, stream, *A, *T, n_wf); break;
        default: hipLaunchKernelGGL(dsp_fir_f16_kernel<2>, grid, dim3(512), lds_bytes, stream, *A, *T, n_wf); break;
    }
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_fir_f16_lds(int lds_bytes) {
    const void* k[3] = {reinterpret_cast<const void*>(&dsp_fir_f16_kernel<0>), reinterpret_cast<const void*>(&dsp_fir_f16_kernel<1>),
                        reinterpret_cast<const void*>(&dsp_fir_f16_kernel<2>)};
    for (int i = 0; i < 3; ++i) {
        const int rc = (int)hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" const char* dsp_internal_fir_f16_kernel_name() { return "dsp_fir_f16_kernel"; }
