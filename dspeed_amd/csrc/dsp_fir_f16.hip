// dsp_fir_f16.hip -- the long FIRs of dsp_fir_mfma.hip (convolve_wf 'v' + numpy.amax, and convolve_wf in any mode with its output kept) on
// the HALF-PRECISION matrix instructions, float32-accurate.
//
// v_mfma_f32_16x16x4_f32 is an exact float32 chain but runs at 1/16 of the rate of v_mfma_f32_16x16x32_f16.  A float32 number is the sum
// of two float16 numbers to 22 bits -- hi = half(x), lo = half(x - hi) -- once it is scaled into float16's range by a power of two, so
//     x . t  =  hi_x hi_t  +  hi_x lo_t  +  lo_x hi_t  (+ lo_x lo_t, 2^-22 of the product: dropped)
// is three float16 products, each exact in the instruction's float32 accumulator: 3/16 of the matrix time of the float32 form.  Measured
// against float64 on the cusp / zac kernels and BASELINE's rows the result is as close as the float32 kernel's (1.5e-7 / 5.5e-7 of the
// filtered waveform's peak, tools/fir_f16_accuracy.py): what limits both is the float32 accumulation, and as there the partial sums leave
// float32 every 256 samples and are added in float64.
//
//   * rows: dsp_fir_f16_rows_kernel streams every row once (a wavefront per row) and leaves its power-of-two scale -- the largest
//     |sample - baseline| lands in [2^14, 2^15) -- and its flags (a NaN, an infinity); the split keeps 22 bits whatever the waveform's
//     magnitude and the scale is undone, exactly, on the way out;
//   * taps: dsp_fir_f16_prep_kernel scales the kernel by one power of two, splits it, reverses it and writes it with zero margins EIGHT
//     times, shifted by 0..7 elements: the B fragment of lane (column c, k-block h) is the 8 consecutive taps kr[k + 8h - c ..], a 16-byte
//     read that is aligned in the copy shifted by (-c) mod 8 -- the Toeplitz matrix is never materialised;
//   * a workgroup = 64 rows x 320 columns (one kernel's 'valid' outputs, or a column tile of a kept output); per stage of 64 samples the
//     64 x 64 A tile (two float16 planes) and the 400-tap window of the 16 tap copies are staged in LDS, double buffered; a wavefront owns
//     2 x 5 tiles of 16 x 16 and issues 30 MFMAs per 32 samples against 14 ds_read_b128; a column tile that meets no tap of a short kernel
//     in a 32-sample group is skipped;
//   * rows holding a NaN / an infinity are handled as by the float32 kernels (all NaN, or tap by tap in float32).
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

#include <cstdlib>
#include <type_traits>

#define FIR_LDS __attribute__((address_space(3)))
#define FIR_GLOBAL __attribute__((address_space(1)))
#define FIR_KARG __attribute__((address_space(4)))

namespace {

#ifndef F16_BK
#define F16_BK 64
#endif
#ifndef F16_BM
#define F16_BM 64
#endif
constexpr int BM = F16_BM, BN = 320, BK = F16_BK, MT = 2, NT = 5;  // BM 64: 8 wavefronts (2 x 4); BM 32: 4 wavefronts, two workgroups per CU
constexpr int NTHR = BM * 8;                                        // 8 threads stage a row
constexpr int TAPV = (16 * ((336 + F16_BK) / 8) + NTHR - 1) / NTHR;  // tap-window vectors a thread stages
constexpr int SV = BK / 64;  // 8-sample vectors a thread stages per stage
// LDS reads are ds_read_b128: four fixed groups of 16 lanes per instruction, 64 banks -- a group is conflict-free when its 16 addresses fall
// into 16 different 16-byte slots of the 256-byte bank line (MI355X_MICROARCH.md, LDS).
constexpr int APITCH = BK + 16;  // halfs per A row: 10 (BK 64) / 18 (BK 128) slots, = 2 modulo 16: lane (row j, k-block h) sits in slot 10 j + h, no two alike in a group
constexpr int TB = 336;          // zero margin below tap 0: window index TB + k - column - e - shift is never negative
constexpr int TWIN = TB + BK;    // taps a stage's fragments can reach
// a tap copy in LDS: a multiple of 256 bytes, so that a copy's slot is its own offset only; copy r starts tap_slot[e][r] slots in -- for
// every alignment e of the window a table that puts the 16 lanes of every group (eight columns x two k-blocks, five copies apart at most
// identical addresses, which broadcast) into 16 different slots (found by search, tools/fir_f16_banks.py; the plain pitch had 51 % of the
// LDS-array cycles as conflicts)
constexpr int TPITCH = ((TWIN + 8 + 15 * 8 + 127) / 128) * 128;
__constant__ unsigned char tap_slot[8][8] = {{2, 5, 9, 6, 15, 12, 9, 15}, {7, 2, 8, 3, 5, 0, 15, 12}, {1, 11, 2, 13, 5, 9, 8, 6}, {8, 4, 11, 1, 4, 0, 15, 8},
                                             {7, 9, 3, 12, 15, 7, 13, 2}, {8, 11, 2, 14, 6, 9, 12, 3}, {3, 7, 6, 11, 0, 4, 15, 11}, {5, 4, 8, 9, 13, 3, 11, 1}};
// F16_PIPE 0: the stages of the amax form as those of the kept-output form (A/B: C3 34.0 M waveforms/s; 35.2 M with one vector instruction dealt
// out per matrix instruction, 35.8 M with two)
#ifndef F16_PIPE
#define F16_PIPE 1
#endif
#ifndef F16_PIPE_VALU
#define F16_PIPE_VALU 2
#endif
#ifndef F16_KFLUSH
#define F16_KFLUSH 256
#endif
constexpr int KFLUSH = F16_KFLUSH;      // samples between two float64 flushes ('valid' + amax over thousands of taps)
constexpr int KFLUSH_STORE = 128;  // ... of the kept-output form: short differentiating kernels cancel, partial sums far above the output

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pow2_inverse(float p) { return __uint_as_float((254u << 23) - __float_as_uint(p)); }  // 1 / 2^k, exactly

// ---- taps: scaled, split, reversed, zero-margined, eight shifted copies.  Layout of `dst` (halfs): [split 0..1][shift 0..7][TZ], then one float
// (the inverse scale) at byte offset 2 * 16 * TZ.  tapz[idx] = kr[idx - TB] for 0 <= idx - TB < m, else 0;  copy_r[i] = tapz[i + r].
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
    if (!(mx > 0.0f) || e < -100 || e > 100) e = 14;           // all zeros (or beyond any sensible kernel): scale 1
    const float scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);  // mx * scale in [2^14, 2^15)
    // (every workgroup finds the scale for itself -- m reads -- and writes its share of the copies: one workgroup alone took 67 us for a
    // 5792-tap kernel, a fifth of a percent of C3 per launch and two launches per pass)
    for (int idx = (int)blockIdx.x * 256 + tid; idx < 8 * TZ; idx += (int)gridDim.x * 256) {
        const int r = idx / TZ, i = idx - r * TZ;
        const int t = i + r - TB;
        const float v = (t >= 0 && t < m) ? taps[m - 1 - t] * scale : 0.0f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        dst[(size_t)r * TZ + i] = hi;
        dst[(size_t)(8 + r) * TZ + i] = lo;
    }
    if (tid == 0 && blockIdx.x == 0) *(float*)(dst + (size_t)16 * TZ) = pow2_inverse(scale);
}

// ---- rows: one wavefront per row; scale[row] = 2^(14 - floor(log2(max |x - baseline|))) over the slice, flags[row] bit 0: an infinity (or a
// magnitude whose scale would leave float32), bit 1: a NaN in the slice or -- bl_subtract's rule, DSP_OP_LOAD ip[0..1] -- around it.
template <int IN>
__global__ void __launch_bounds__(256) dsp_fir_f16_rows_kernel(FirArgs A_, float* __restrict__ scale, unsigned* __restrict__ flags, int64_t n_wf) {
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    const int lane = (int)threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (row >= n_wf) return;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const int n = A.n;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (row * A.wf_stride + A.wf_offset) * ESZ;
    const bool sub = A.sub_mode != 0;
    const float bl = sub ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[row * A.bl_stride] : A.bl_const) : 0.0f;
    auto at = [&](int i) -> float {
        const float x = IN == 0 ? ((const FIR_GLOBAL float*)rowp)[i] : (IN == 1 ? (float)((const FIR_GLOBAL short*)rowp)[i] : (float)((const FIR_GLOBAL unsigned short*)rowp)[i]);
        return sub ? x - bl : x;
    };
    float mx = 0.0f;
    bool has_nan = false;
    auto see = [&](float x) {
        const float a = __builtin_fabsf(x);
        has_nan |= (x != x);
        mx = a > mx ? a : mx;  // (a NaN never raises it)
    };
    const int n8 = n & ~7;
    for (int i0 = lane * 8; i0 < n8; i0 += 4 * 512) {  // 8 samples per lane and load, four loads in flight
        float x[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + b * 512;
            if (i < n8) {
                if (IN == 0) {
#ifdef F16_ROWS_PLAIN_LOADS
                    const f4 v0 = *(const FIR_GLOBAL f4*)(rowp + (size_t)i * 4), v1 = *(const FIR_GLOBAL f4*)(rowp + (size_t)i * 4 + 16);
#else
                    const f4 v0 = __builtin_nontemporal_load((const FIR_GLOBAL f4*)(rowp + (size_t)i * 4));
                    const f4 v1 = __builtin_nontemporal_load((const FIR_GLOBAL f4*)(rowp + (size_t)i * 4 + 16));
#endif
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[b][u] = v0[u];
                        x[b][4 + u] = v1[u];
                    }
                } else {
#ifdef F16_ROWS_PLAIN_LOADS
                    const u4 raw = *(const FIR_GLOBAL u4*)(rowp + (size_t)i * 2);
#else
                    const u4 raw = __builtin_nontemporal_load((const FIR_GLOBAL u4*)(rowp + (size_t)i * 2));
#endif
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[b][2 * u] = IN == 1 ? (float)(short)(raw[u] & 0xffffu) : (float)(raw[u] & 0xffffu);
                        x[b][2 * u + 1] = IN == 1 ? (float)(short)(raw[u] >> 16) : (float)(raw[u] >> 16);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (i0 + b * 512 < n8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) see(sub ? x[b][u] - bl : x[b][u]);
            }
    }
    for (int i = n8 + lane; i < n; i += 64) see(at(i));
    if (IN == 0 && sub) {
        for (int i = -A.scan_before + lane; i < 0; i += 64) has_nan |= (at(i) != at(i));
        for (int i = n + lane; i < n + A.scan_after; i += 64) has_nan |= (at(i) != at(i));
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
    has_nan = __any(has_nan);
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    const bool bad = !(mx <= 3.4028234663852886e38f) || (mx > 0.0f && (e < -100 || e > 100));
    if (!(mx > 0.0f) || bad) e = 14;
    if (lane == 0) {
        scale[row] = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        flags[row] = (bad ? 1u : 0u) | (has_nan ? 2u : 0u);
    }
}

// STORE: a 320-column tile of a kept output; else kernel q's 'valid' outputs and their maximum (which tile / kernel and which block of rows: from the workgroup id, below)
// RES (kept output, short kernels): the tap copies a tile can reach -- kt + TB taps each -- fit the room of the two window buffers, so they are
// staged once, before the first stage, and no stage fetches or writes taps (launch_f16 decides: the longest window of a tile must fit)
constexpr int RES_HALFS = 2 * 16 * TPITCH;  // halfs of the tap region
__host__ __device__ constexpr int res_pitch(int kt) { return ((kt + TB + 8 + 15 * 8 + 127) / 128) * 128; }

template <int IN, bool STORE, bool RES = false>
__global__ void __launch_bounds__(NTHR, 1) dsp_fir_f16_kernel(FirArgs A_, FirF16Taps T_, int64_t n_wf, int parts) {
    static_assert(!RES || STORE, "resident taps: the kept-output form");
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) unsigned char f16_smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BM == 64 ? (wave & 1) : 0, wn = BM == 64 ? (wave >> 1) : wave;  // rows 32 wm ..; column tiles wn, wn + 4, .. (16 columns each)
    // Which (kernel or column tile, block of 64 rows) this workgroup is.  Workgroup ids go round the 8 XCDs, each with an L2 of its own: every
    // workgroup that reads one block of rows -- the kernels of the amax form, the column tiles of a kept output, whose windows overlap -- gets
    // ids 8 apart, i.e. the same XCD one after the other, and all but the first find the rows in that L2 (id = 8 * slot + xcd; slot = (block
    // group, part), row block = 8 * group + xcd).  F16_PLAIN_GRID: id = part + parts * row block, as before, for the A/B.
    // (parts: kernels / column tiles)
#ifdef F16_PLAIN_GRID
    const int part = (int)(blockIdx.x % (unsigned)parts);
    const int64_t rblock = blockIdx.x / (unsigned)parts;
#else
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    const int part = slot % parts;
    const int64_t rblock = (int64_t)(slot / parts) * 8 + xcd;
#endif
    if (rblock * BM >= n_wf) return;  // (the grid is padded to whole groups of 8 row blocks; whole workgroups leave, before any barrier)
    const int q = STORE ? 0 : part;
    const int64_t row0 = rblock * BM;
    const int n = A.n, m = A.m[q], P = A.p[q];
    // the K window of this workgroup: samples ks .. ks + kt - 1 of the rows; output column cl sums window samples cl + e .. cl + e + m - 1
    const int c0 = STORE ? part * BN : 0;
    const int s0 = c0 - (STORE ? A.dshift : 0);
    const int ks = (s0 >> 3) << 3, e = s0 - ks;  // (arithmetic shift: rounds down for the negative start of the first tiles)
    const int cols = P - c0 < BN ? P - c0 : BN;
    const int kt = STORE ? ((cols + m - 1 + e + BK - 1) / BK) * BK : ((A.kend + BK - 1) / BK) * BK;
    const int TZ = T_.tz;
    const FIR_GLOBAL _Float16* tg = (const FIR_GLOBAL _Float16*)T_.taps16[q];
    const float tap_inv = *(const FIR_GLOBAL float*)(tg + (size_t)16 * TZ);

    // LDS: A planes [buf][split][BM][APITCH], tap windows [buf][split * 8 + shift][TPITCH], reduction scratch
    FIR_LDS _Float16* As = (FIR_LDS _Float16*)f16_smem;
    FIR_LDS _Float16* Tw = As + 2 * 2 * BM * APITCH;
    FIR_LDS float* red = (FIR_LDS float*)(Tw + 2 * 16 * TPITCH);  // [BM][4][2]
    FIR_LDS float* rback = red + BM * 4 * 2;                       // [BM]: what undoes the row's and the kernel's scales

    // ---- staging geometry: thread t carries 8 consecutive samples of row t / 8
    const int srow = tid >> 3, skc = (tid & 7) * 8;
    const int64_t grow = row0 + srow < n_wf ? row0 + srow : n_wf - 1;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (grow * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[grow * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    const float xs = ((const FIR_GLOBAL float*)T_.row_scale)[grow];
    if ((tid & 7) == 0) rback[srow] = pow2_inverse(xs) * tap_inv;

    // Two register sets: the samples (and taps) of stage s + 2 are requested while stage s is multiplied and stage s + 1 waits in the other set
    // for its turn to be converted into LDS -- a request has two stages' worth of matrix work to come back (one was not enough: a quarter of
    // the kernel's time went into waiting for it)
    // (a request's registers are not touched before `commit`: anything that reads them -- a conversion, the baseline -- makes the compiler wait
    // for the data right behind the request; and the requests are unconditional, from a clamped address, because a branch around a load makes
    // the wait-count insertion give up and wait for everything at the join)
    struct Raw {
        u4 w[IN == 0 ? 2 : 1];  // 8 samples as they lie in the row: 32 bytes of float32 or 16 of int16 / uint16
    };
    Raw stage_xx[2][SV];
    auto fetch = [&](int k0, Raw (&raw)[SV]) {
#pragma unroll
        for (int v = 0; v < SV; ++v) {
            int i = ks + k0 + 64 * v + skc;  // (a multiple of 8: a vector lies below sample 0 whole or not at all)
            i = i < 0 ? 0 : (i >= n ? ((n - 1) & ~7) : i);  // outside the slice: any vector of the row (`commit` puts zeros there)
            const FIR_GLOBAL u4* src = (const FIR_GLOBAL u4*)(rowp + (size_t)i * ESZ);
            raw[v].w[0] = src[0];
            if (IN == 0) raw[v].w[IN == 0 ? 1 : 0] = src[1];
        }
    };
    // zeros outside the slice, baseline, scale, split, into the two planes.  MASK: the stage reaches outside the slice (its first or last one, a
    // scalar test for the whole workgroup); every other stage skips the per-sample selects, 24 of its 60 vector instructions per 8 samples
    struct Split {
        h8 hi, lo;
    };
    auto convert_as = [&](auto mask, int k0, const Raw (&raw)[SV], Split (&out)[SV]) {
        constexpr bool MASK = decltype(mask)::value;
#pragma unroll
        for (int v = 0; v < SV; ++v) {
            const int i = ks + k0 + 64 * v + skc;
            const int live = i < 0 ? 0 : n - i;  // samples of this vector inside the slice (>= 8 almost always; the host keeps the row readable to the vector's end)
            float x[8];
            if (IN == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x[u] = __uint_as_float(raw[v].w[0][u]);
                    x[4 + u] = __uint_as_float(raw[v].w[IN == 0 ? 1 : 0][u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned r = raw[v].w[0][u];
                    x[2 * u] = IN == 1 ? (float)(short)(r & 0xffffu) : (float)(r & 0xffffu);
                    x[2 * u + 1] = IN == 1 ? (float)(short)(r >> 16) : (float)(r >> 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = sub ? x[u] - bl : x[u];
                if (MASK) x[u] = u < live ? x[u] : 0.0f;
            }
            h8 hi, lo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float y = x[u] * xs;
                const _Float16 h = (_Float16)y;
                hi[u] = h;
                lo[u] = (_Float16)(y - (float)h);
            }
            out[v].hi = hi;
            out[v].lo = lo;
        }
    };
    auto put = [&](int buf, const Split (&sp)[SV]) {
#pragma unroll
        for (int v = 0; v < SV; ++v) {
            FIR_LDS _Float16* ap = As + buf * 2 * BM * APITCH + srow * APITCH + 64 * v + skc;
            *(FIR_LDS h8*)ap = sp[v].hi;
            *(FIR_LDS h8*)(ap + BM * APITCH) = sp[v].lo;
        }
    };
    auto commit_as = [&](auto mask, int buf, int k0, const Raw (&raw)[SV]) {
        Split sp[SV];
        convert_as(mask, k0, raw, sp);
        put(buf, sp);
    };
    auto commit = [&](int buf, int k0, const Raw (&raw)[SV]) {
        if (ks + k0 >= 0 && ks + k0 + BK <= n)  // (uniform)
            commit_as(std::false_type(), buf, k0, raw);
        else
            commit_as(std::true_type(), buf, k0, raw);
    };
    // the 16 tap copies' windows [k0, k0 + TWIN): 16 x 50 vectors of 8 halfs over the workgroup's threads (a thread beyond the last vector
    // requests the last one again and does not store it)
    h8 tap_vv[2][TAPV];
    int tap_src[TAPV], tap_dst[TAPV];  // (this thread's vectors: where they come from in the image, where they go in the window)
    bool tap_live[TAPV];
#pragma unroll
    for (int it = 0; it < TAPV; ++it) {
        const int v = tid + it * NTHR;
        tap_live[it] = v < 16 * (TWIN / 8);
        const int vc = tap_live[it] ? v : 16 * (TWIN / 8) - 1;
        const int c = vc / (TWIN / 8), o = (vc - c * (TWIN / 8)) * 8;
        tap_src[it] = c * TZ + o;
        tap_dst[it] = c * TPITCH + o + 8 * tap_slot[e][c & 7];
    }
    const int rpitch = res_pitch(kt);
    if constexpr (RES) {  // all of it, once: 16 copies x (kt + TB) taps
        const int per = (kt + TB) / 8, total = 16 * per;
        for (int v0 = tid; v0 < total; v0 += 4 * NTHR) {
            h8 tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * NTHR < total ? v0 + u * NTHR : total - 1;
                const int c = v / per, o = (v - c * per) * 8;
                tv[u] = *(const FIR_GLOBAL h8*)(tg + c * TZ + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * NTHR;
                if (v < total) {
                    const int c = v / per, o = (v - c * per) * 8;
                    *(FIR_LDS h8*)(Tw + c * rpitch + o + 8 * tap_slot[e][c & 7]) = tv[u];
                }
            }
        }
    }
    auto fetch_taps = [&](int k0, h8 (&tap_v)[TAPV]) {
#pragma unroll
        for (int it = 0; it < TAPV; ++it) tap_v[it] = *(const FIR_GLOBAL h8*)(tg + tap_src[it] + k0);
    };
    auto commit_taps = [&](int buf, const h8 (&tap_v)[TAPV]) {
#pragma unroll
        for (int it = 0; it < TAPV; ++it)
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
    // B fragment of column cl = 16 (wn + 4 tn) + j: tapz[TB + kk - cl - e ..] for window samples kk = k0 + 32 g + 8 h4 ..: aligned in the copy
    // shifted by r = -(j + e) mod 8, at window offset TB + 32 g + 8 h4 - cl - e - r
    const int shift = (8 - ((j + e) & 7)) & 7;
    const int t_pitch = RES ? rpitch : TPITCH;
    const int t_off = shift * t_pitch + 8 * tap_slot[e][shift] + TB + 8 * h4 - (wn * 16 + j) - e - shift;

    const int n_stage = kt / BK;
    fetch(0, stage_xx[0]);
    if constexpr (!RES) fetch_taps(0, tap_vv[0]);
    if (n_stage > 1) {
        fetch(BK, stage_xx[1]);
        if constexpr (!RES) fetch_taps(BK, tap_vv[1]);
    }
    commit(0, 0, stage_xx[0]);
    if constexpr (!RES) commit_taps(0, tap_vv[0]);
    __syncthreads();
    // stage st (parity PAR): its operands are in LDS buffer PAR; register set 1 - PAR holds stage st + 1, register set PAR is free for st + 2
    // PIPE (a stage well inside the rows and the run of stages: nothing conditional in it): the operands of the next stage are converted and written
    // BEFORE this stage's matrix instructions in program order and the scheduler is told to deal the two out alternately -- a wavefront issues its
    // conversions in the shadow of its own matrix instructions instead of everybody converting while the matrix pipe idles
    auto stage = [&](auto par, auto pipe, int st) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool PIPE = decltype(pipe)::value;
        const int k0 = st * BK;
        Split piped[SV];
        if constexpr (PIPE) {
            fetch(k0 + 2 * BK, stage_xx[PAR]);
            if constexpr (!RES) fetch_taps(k0 + 2 * BK, tap_vv[PAR]);
            convert_as(std::false_type(), k0 + BK, stage_xx[1 - PAR], piped);
        }
#ifndef F16_DIAG_NO_FETCH
        if (!PIPE && st + 2 < n_stage) {
            fetch(k0 + 2 * BK, stage_xx[PAR]);
            if constexpr (!RES) fetch_taps(k0 + 2 * BK, tap_vv[PAR]);
        }
#endif
        const FIR_LDS _Float16* ab = As + PAR * 2 * BM * APITCH + a_off;
        const FIR_LDS _Float16* tb = RES ? Tw + t_off + k0 : Tw + PAR * 16 * TPITCH + t_off;
#pragma unroll
        for (int g = 0; g < BK / 32; ++g) {
            const int kb = k0 + g * 32;  // this group: window samples kb .. kb + 31
            h8 ah[MT], al[MT];
#pragma unroll
            for (int tm = 0; tm < MT; ++tm) {
#ifdef F16_DIAG_NO_LDSREAD
                ah[tm] = h8{(_Float16)(float)kb, 0, 0, 0, 0, 0, 0, 0};
                al[tm] = h8{(_Float16)(float)lane, 0, 0, 0, 0, 0, 0, 0};
#else
                ah[tm] = *(const FIR_LDS h8*)(ab + tm * 16 * APITCH + g * 32);
                al[tm] = *(const FIR_LDS h8*)(ab + BM * APITCH + tm * 16 * APITCH + g * 32);
#endif
            }
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                // (uniform: a column tile meets a 32-sample group only where a tap lies -- window sample kk meets column cl at tap kk - cl - e)
                const int c_lo = 16 * (wn + 4 * tn) + e;
                if (!STORE || (kb + 31 >= c_lo && kb <= c_lo + 14 + m)) {
#ifdef F16_DIAG_NO_LDSREAD
                    const h8 bh = h8{(_Float16)(float)(kb + tn), 0, 0, 0, 0, 0, 0, 0}, blo = h8{(_Float16)(float)(lane + tn), 0, 0, 0, 0, 0, 0, 0};
#else
                    const h8 bh = *(const FIR_LDS h8*)(tb + g * 32 - tn * 64);
                    const h8 blo = *(const FIR_LDS h8*)(tb + 8 * t_pitch + g * 32 - tn * 64);
#endif
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], blo, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bh, acc[tm][tn], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (PIPE) {
            put(1 - PAR, piped);
            if constexpr (!RES) commit_taps(1 - PAR, tap_vv[1 - PAR]);
#pragma unroll
            for (int i = 0; i < (BK / 32) * NT * MT * 3; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one matrix instruction
                __builtin_amdgcn_sched_group_barrier(0x002, F16_PIPE_VALU, 0);  // vector instructions of the conversion behind it
            }
        }
#ifndef F16_DIAG_NO_COMMIT
        if (!PIPE && st + 1 < n_stage) {
            commit(1 - PAR, k0 + BK, stage_xx[1 - PAR]);
            if constexpr (!RES) commit_taps(1 - PAR, tap_vv[1 - PAR]);
        }
#endif
#ifndef F16_DIAG_NO_BARRIER
        // (not __syncthreads(): the compiler lowers that to vmcnt(0) as well, and the requests of the stage after next must stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    };
    // (two loops: the float64 flush sits between runs of stages, outside the stage loop -- inside it, as a conditional, the compiler copies
    // all 120 accumulator registers around the branch in every stage; a run is an even number of stages, so the parity is static)
    constexpr int RUN = (STORE ? KFLUSH_STORE : KFLUSH) / BK;
    static_assert(RUN % 2 == 0 || RUN == 1, "a run of stages keeps the buffer parity");
    // (a kept output of a short kernel flushed once at the end instead -- a column's sum spans m samples only: 3.76 -> 3.56 ms for the
    // recipe's 133-tap filter, errors 5.5e-7 -> 6.4e-7 of the peak and 4.6e-7 -> 9.5e-7 for 250 taps, tools/fir_f16_store_accuracy.py: left)
    constexpr int run = RUN;
    auto flush = [&]() {  // partial sums of (up to) 256 / 128 samples leave float32 here
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tot[tm][tn][r] += (double)acc[tm][tn][r];
                acc[tm][tn] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
    };
    // stages 0 .. n_pipe - 1: whole runs whose every stage has two more behind it and converts samples inside the rows only
    int n_pipe = 0;
#if F16_PIPE
    if (!STORE && RUN > 1) {
        int last = n_stage - 3;                        // st + 2 < n_stage
        const int inside = (n - ks) / BK - 2;          // ks + (st + 2) BK <= n
        last = last < inside ? last : inside;
        n_pipe = ks >= 0 && last >= 0 ? ((last + 1) / RUN) * RUN : 0;
    }
#endif
    for (int st0 = 0; st0 < n_pipe; st0 += RUN) {
        for (int st = st0; st < st0 + RUN; st += 2) {
            stage(std::integral_constant<int, 0>(), std::true_type(), st);
            stage(std::integral_constant<int, 1>(), std::true_type(), st + 1);
        }
        flush();
    }
    for (int st0 = n_pipe; st0 < n_stage; st0 += run) {
        const int st1 = st0 + run < n_stage ? st0 + run : n_stage;
        if (RUN == 1) {
            if (st0 & 1) stage(std::integral_constant<int, 1>(), std::false_type(), st0); else stage(std::integral_constant<int, 0>(), std::false_type(), st0);
        } else {
            for (int st = st0; st < st1; st += 2) {
                stage(std::integral_constant<int, 0>(), std::false_type(), st);
                if (st + 1 < st1) stage(std::integral_constant<int, 1>(), std::false_type(), st + 1);
            }
        }
        flush();
    }
    // C layout of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
    if (STORE) {
        FIR_GLOBAL float* outp = (FIR_GLOBAL float*)A.out[0];
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 32 + tm * 16 + h4 * 4 + r;
                const int64_t row = row0 + rl;
                const double back = (double)rback[rl];  // (a power of two: exact)
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    const int cl = 16 * (wn + 4 * tn) + j;
                    if (row < n_wf && cl < cols) outp[row * A.out_stride[0] + c0 + cl] = (float)(tot[tm][tn][r] * back);
                }
            }
        return;  // (rows with a NaN or an infinity: dsp_fir_fixup_kernel, launched behind this one)
    }
    // ---- numpy.amax over the valid columns of every row
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = wm * 32 + tm * 16 + h4 * 4 + r;
            const double back = (double)rback[rl];
            float vmax = -__builtin_inff();
            bool vnan = false;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const float v = (float)(tot[tm][tn][r] * back);
                const bool valid = 16 * (wn + 4 * tn) + j < P;
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
    __syncthreads();
    FIR_LDS unsigned* rowflag = (FIR_LDS unsigned*)(As);  // (the A buffers are free now)
    if (tid < BM) rowflag[tid] = row0 + tid < n_wf ? ((const FIR_GLOBAL unsigned*)T_.row_flags)[row0 + tid] : 0u;
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
        for (int jo = tid; jo < P; jo += NTHR) {
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
            for (int w = 1; w < NTHR / 64; ++w) {
                vmax = fmaxf(vmax, red[w * 2]);
                fn = fmaxf(fn, red[w * 2 + 1]);
            }
            outp[(row0 + rl) * A.out_stride[q]] = fn != 0.0f ? quiet_nan<float>() : vmax;
        }
    }
}

}  // namespace

// halfs of the tap image of one kernel: 16 copies of TZ = the longest K window rounded up to a stage + the window the last stage reaches +
// the margin the shifted copies reach into, then the inverse scale (one float, kept 16-byte aligned)
extern "C" int dsp_internal_fir_f16_tz(int kend) { return ((kend + 8 + BK - 1) / BK) * BK + TWIN + 16; }
extern "C" size_t dsp_internal_fir_f16_taps_bytes(int kend) { return (size_t)16 * dsp_internal_fir_f16_tz(kend) * 2 + 16; }
extern "C" int dsp_internal_fir_f16_lds_bytes() { return (2 * 2 * BM * APITCH + 2 * 16 * TPITCH) * 2 + (BM * 4 * 2 + BM) * 4; }

extern "C" int dsp_internal_fir_fixup(const FirArgs* A, int64_t n_wf, hipStream_t stream);  // dsp_fir_mfma.hip

template <int IN>
static void launch_f16(const FirArgs* A, const FirF16Taps* T, int64_t n_wf, int lds_bytes, hipStream_t stream) {
    if (!T->rows_done)
        hipLaunchKernelGGL(dsp_fir_f16_rows_kernel<IN>, dim3((unsigned)((n_wf + 3) / 4)), dim3(256), 0, stream, *A, (float*)T->row_scale,
                           (unsigned*)T->row_flags, n_wf);
    int parts;
    const int64_t rblocks = (n_wf + BM - 1) / BM, rgroups = (rblocks + 7) / 8;
    if (A->store) {
        parts = (A->p[0] + BN - 1) / BN;
        const dim3 grid((unsigned)(8 * rgroups * parts));
        // the longest window of a tile: its columns, the kernel, the alignment of its first sample (e <= 7), in whole stages
        const int cols = A->p[0] < BN ? A->p[0] : BN, kt_max = ((cols + A->m[0] - 1 + 7 + BK - 1) / BK) * BK;
        static const bool no_res = getenv("DSPEED_HIP_FIR_NO_RESIDENT_TAPS") && getenv("DSPEED_HIP_FIR_NO_RESIDENT_TAPS")[0] == '1';
        if (!no_res && 16 * res_pitch(kt_max) <= RES_HALFS)
            hipLaunchKernelGGL((dsp_fir_f16_kernel<IN, true, true>), grid, dim3(NTHR), lds_bytes, stream, *A, *T, n_wf, parts);
        else
            hipLaunchKernelGGL((dsp_fir_f16_kernel<IN, true>), grid, dim3(NTHR), lds_bytes, stream, *A, *T, n_wf, parts);
    } else {
        parts = A->n_kernels;
        const dim3 grid((unsigned)(8 * rgroups * parts));
        hipLaunchKernelGGL((dsp_fir_f16_kernel<IN, false>), grid, dim3(NTHR), lds_bytes, stream, *A, *T, n_wf, parts);
    }
}

extern "C" int dsp_internal_launch_fir_f16(const FirArgs* A, const FirF16Taps* T, int64_t n_wf, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0 || A->n_kernels <= 0) return 0;
    for (int q = 0; q < A->n_kernels; ++q)
        hipLaunchKernelGGL(dsp_fir_f16_prep_kernel, dim3(32), dim3(256), 0, stream, A->taps[q], A->m[q], T->tz, (_Float16*)T->taps16[q]);
    switch (A->in_kind) {
        case 0: launch_f16<0>(A, T, n_wf, lds_bytes, stream); break;
        case 1: launch_f16<1>(A, T, n_wf, lds_bytes, stream); break;
        default: launch_f16<2>(A, T, n_wf, lds_bytes, stream); break;
    }
    int rc = (int)hipGetLastError();
    if (rc == 0 && A->store) {
        FirArgs F = *A;
        F.row_flags = (const uint32_t*)T->row_flags;  // (the rows are screened: the pass behind reads a word per row, not the row)
        rc = dsp_internal_fir_fixup(&F, n_wf, stream);
    }
    return rc;
}

extern "C" int dsp_internal_set_fir_f16_lds(int lds_bytes) {
    const void* k[9] = {reinterpret_cast<const void*>(&dsp_fir_f16_kernel<0, false>), reinterpret_cast<const void*>(&dsp_fir_f16_kernel<1, false>),
                        reinterpret_cast<const void*>(&dsp_fir_f16_kernel<2, false>), reinterpret_cast<const void*>(&dsp_fir_f16_kernel<0, true>),
                        reinterpret_cast<const void*>(&dsp_fir_f16_kernel<1, true>), reinterpret_cast<const void*>(&dsp_fir_f16_kernel<2, true>),
                        reinterpret_cast<const void*>(&dsp_fir_f16_kernel<0, true, true>), reinterpret_cast<const void*>(&dsp_fir_f16_kernel<1, true, true>),
                        reinterpret_cast<const void*>(&dsp_fir_f16_kernel<2, true, true>)};
    for (int i = 0; i < 9; ++i) {
        const int rc = (int)hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" const char* dsp_internal_fir_f16_kernel_name() { return "dsp_fir_f16_kernel"; }
