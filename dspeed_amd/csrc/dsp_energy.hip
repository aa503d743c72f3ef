// dsp_energy.hip -- the Ge energy chain as ONE specialised kernel (BASELINE.json configs[1]/[3]):
//
//     waveform --bl_subtract--> --pole_zero--> --trap_filter|trap_norm|asym_trap--> fixed_time_pickoff --> 1 float
//
// Same arithmetic as the generic waveform VM (dsp_vm.hip) -- the host selects this kernel when a chain has exactly
// this shape -- but organised for the HBM roofline:
//   * the next waveform's 16 KB are already in flight (16-byte global loads into registers) while the current one
//     is being filtered, so HBM latency is hidden behind the arithmetic of the same wavefront;
//   * bl_subtract is folded into the pole-zero passes, the float64 prefix sums the trapezoid needs are produced by
//     the pole-zero pass itself, and the trapezoid output is never stored (only the picked-off samples are kept):
//     per waveform the LDS sees 2 writes and 6 reads per sample instead of 4 and 9;
//   * every chunk loop is software-pipelined in groups of 8 samples (the next group's LDS reads are issued before
//     the current group's dependent arithmetic).
// One wavefront per waveform, lane j owns samples [jC, (j+1)C), LDS pitch C+1, zero guard of 2 pitches below the slot
// (layout identical to a VM slot).  Reference bodies: processors/bl_subtract.py:11-46, pole_zero.py:24-77,
// trap_filters.py:12-227, fixed_time_pickoff.py:12-125.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

struct EnergyArgs {
    const void* wf;        // waveform rows
    int64_t wf_stride;     // elements between rows
    int32_t wf_offset;     // first sample used
    int32_t len;           // samples per waveform
    const float* bl;       // per-waveform baseline column, or nullptr
    int64_t bl_stride;
    float bl_const;
    int32_t has_bl;        // 0: the chain has no bl_subtract (bl_const is then 0: x - 0 == x exactly)
    const float* tp;       // per-waveform pick-off time column, or nullptr
    int64_t tp_stride;
    float tp_const;
    int32_t mode;          // pick-off mode char
    float* out;
    int64_t out_stride;
    double c;              // exp(-1/tau)
    double rr, ll;         // rise, fall as float64
    int32_t tau_nan;
    int32_t all_nan;       // trap_filter with rise == 0
    int32_t C, pitch;      // samples per lane, C + 1
    float invC;
    int32_t q[3], rho[3];  // lag = q*C + rho
    int32_t lds_elems_per_wave;
    int32_t slot_off;      // element offset of the slot inside the wave's region (2*pitch guard below it)
    int32_t ablate;        // timing experiments only (DSPEED_HIP_ABLATE): bit 0/1/2 = skip pass 1/2/3; results are then wrong
};

namespace {

constexpr int G = 8;   // samples per software-pipeline group in the pole-zero passes
constexpr int G3 = 8;  // ... in the trapezoid replay (4 streams x 2 buffers live there: 64 VGPRs, fine at 2 waves/SIMD)

template <int N>
__device__ __forceinline__ void load_group(float (&v)[N], const float* p) {
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = p[u];
}

template <int NPF, int KIND>
__global__ void __launch_bounds__(256, 2) dsp_energy_kernel(EnergyArgs A, int64_t n_wf, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);  // provably wave-uniform
    float* lds = reinterpret_cast<float*>(smem_raw) + (size_t)wave * A.lds_elems_per_wave;
    for (int e = lane; e < A.lds_elems_per_wave; e += 64) lds[e] = 0.0f;
    wave_sync();

    // the host selects this kernel only for len == 256 * NPF: every lane owns exactly C = 4 * NPF samples, no tail
    constexpr int C = 4 * NPF, pitch = C + 1, len = 256 * NPF;
    float* slot = lds + A.slot_off;
    float* mine = slot + lane * pitch;

    // lagged-read bases (identical for every row): see trap_core in dsp_vm.hip
    const float* lagp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int jj = lane - A.q[k] - 1;
        lagp[k] = (jj >= -1) ? slot + jj * pitch + (C - A.rho[k]) : slot - 2 * pitch;
    }

    const int64_t stride_rows = (int64_t)gridDim.x * wpb;
    int64_t row = (int64_t)blockIdx.x * wpb + wave;

    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 pf[NPF];
    float pf_bl = 0.0f, pf_tp = 0.0f;
    auto prefetch = [&](int64_t r) {
        const float* g = (const float*)A.wf + r * A.wf_stride + A.wf_offset;
#pragma unroll
        for (int b = 0; b < NPF; ++b) pf[b] = reinterpret_cast<const f4*>(g)[b * 64 + lane];
        pf_bl = A.bl ? A.bl[r * A.bl_stride] : A.bl_const;  // 0 when the chain has no bl_subtract: x - 0 == x exactly
        pf_tp = A.tp ? A.tp[r * A.tp_stride] : A.tp_const;
    };
    auto report = [&](int code, int64_t r) {
        if (lane == 0 && atomicCAS(&err[0], 0, code) == 0) {
            err[1] = (int)(r & 0xffffffffll);
            err[2] = (int)(r >> 32);
        }
    };
    if (row < n_wf) prefetch(row);

    for (; row < n_wf; row += stride_rows) {
        // ---- stage the prefetched waveform into LDS (chunked layout)
#pragma unroll
        for (int b = 0; b < NPF; ++b) {
            const int e = (b * 64 + lane) * 4;
            float* d = slot + e + e / C;  // sample e -> element e + e / C (chunk pad)
#pragma unroll
            for (int m = 0; m < 4; ++m) d[m] = pf[b][m];
        }
        // per-waveform scalars are wave-uniform: say so, or every use downstream becomes per-lane (exec-masked) code
        const float bl = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pf_bl)));
        const float t_in = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pf_tp)));
        const int64_t next = row + stride_rows;
        __builtin_amdgcn_sched_barrier(0);  // the staging stores must issue before the registers are reloaded
        if (next < n_wf) prefetch(next);    // in flight while this waveform is filtered
        __builtin_amdgcn_sched_barrier(0);
        wave_sync();

        float result = quiet_nan<float>();
        // ---- pass 1: per-chunk float64 sum of x = w - baseline; a NaN anywhere (or a NaN baseline) poisons the sum
        double X = 0.0;
        if (!(A.ablate & 1)) {
            float va[G], vb[G];
            load_group(va, mine);
#pragma unroll 1
            for (int t = 0; t < C; t += 2 * G) {
                load_group(vb, mine + t + G);
#pragma unroll
                for (int u = 0; u < G; ++u) X += (double)(va[u] - bl);
                if (t + 2 * G < C) load_group(va, mine + t + 2 * G);
#pragma unroll
                for (int u = 0; u < G; ++u) X += (double)(vb[u] - bl);
            }
        }
        bool in_nan = A.tau_nan != 0;
        if (wave_any(!(fabs(X) <= 1.7976931348623157e308))) {
            // NaN or infinite sum: look for real NaNs (an infinite input is not NaN for the reference, pole_zero.py:55-58)
            bool n = false;
            for (int t = 0; t < C; ++t) {
                const float x = mine[t] - bl;
                n |= (x != x);
            }
            in_nan |= wave_any(n);
        }
        if (!in_nan) {
            const double E = wave_exscan_add(X);
            const float xlast = mine[C - 1] - bl;
            const double xprev0 = (double)wave_prev(xlast);
            // ---- pass 2: pole-zero recurrence in the reference's operation order, output in place; float32 running sum of
            // the output feeds the speculative carries of the trapezoid
            const double c = A.c;
            double acc = E - c * (E - xprev0), xp = xprev0;
            float run = 0.0f, cap[3] = {0.0f, 0.0f, 0.0f};
            if (!(A.ablate & 2)) {
                float va[G], vb[G];
                load_group(va, mine);
                auto body = [&](float (&v)[G], int t) {
                    float rs[G];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const double x = (double)(v[u] - bl);
                        acc = (acc + x) - xp * c;
                        const float y = (float)acc;
                        mine[t + u] = y;
                        xp = x;
                        run += y;
                        rs[u] = run;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int d = ((C - A.rho[k]) % C) - t;  // prefix needed by lag k completes after d samples of this group
                        if (d >= 1 && d <= G) {
#pragma unroll
                            for (int u = 0; u < G; ++u)
                                if (d == u + 1) cap[k] = rs[u];
                        }
                    }
                };
#pragma unroll 1
                for (int t = 0; t < C; t += 2 * G) {
                    load_group(vb, mine + t + G);
                    body(va, t);
                    if (t + 2 * G < C) load_group(va, mine + t + 2 * G);
                    body(vb, t + G);
                }
            }
            wave_sync();
            bool pz_nan = false;
            if (wave_any(!(fabsf(run) <= 3.4028234663852886e38f))) {
                bool n = false;
                for (int t = 0; t < C; ++t) {
                    const float y = mine[t];
                    n |= (y != y);
                }
                pz_nan = wave_any(n);
            }
            if (pz_nan) {
                report(DSP_E_PZ_NAN, row);  // pole_zero.py:76-77
            } else if (!A.all_nan && !(A.ablate & 4) && pickoff_in_range(t_in, len)) {
                // ---- speculative carries: filter value at every chunk boundary from the prefix sums
                const double Ep = wave_exscan_add((double)run);
                double Ak[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool whole = (A.rho[k] == 0);
                    Ak[k] = wave_shift_up(Ep + (whole ? 0.0 : (double)cap[k]), A.q[k] + (whole ? 0 : 1));
                }
                double Gd;
                if (KIND == TRAP_FILTER)
                    Gd = ((Ep - Ak[0]) - Ak[1]) + Ak[2];
                else if (KIND == TRAP_NORM)
                    Gd = (((Ep - Ak[0]) - Ak[1]) + Ak[2]) / A.rr;
                else
                    Gd = (Ep - Ak[0]) / A.rr - (Ak[1] - Ak[2]) / A.ll;
                const float g = (lane == 0) ? -0.0f : (float)Gd;

                // ---- pick-off positions (uniform): samples i0-1 .. i0+2, kept only where the mode needs them
                const int i0 = (int)t_in;
                const bool wide = (A.mode == 'h');
                int cl[4], co[4];
                float capv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = i0 - 1 + k;
                    const bool need = ((k == 1) || (k == 2) || wide) && e >= 0 && e < len;
                    const int l = need ? e / C : -1;
                    cl[k] = l;
                    co[k] = need ? e - l * C : -1000;
                    capv[k] = 0.0f;
                }
                // groups of the replay loop that contain a wanted sample (uniform bit mask)
                unsigned capmask = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (co[k] >= 0) capmask |= 1u << (co[k] / G3);
                // ---- pass 3: replay the reference's float32 rounding sequence over the chunk
                float y = g;
                {
                    float a0[G3], a1[G3], a2[G3], a3[G3], b0[G3], b1[G3], b2[G3], b3[G3];
                    auto fetch = [&](float (&o)[G3], float (&l0)[G3], float (&l1)[G3], float (&l2)[G3], int t) {
                        load_group(o, mine + t);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            float(&dst)[G3] = (k == 0) ? l0 : (k == 1 ? l1 : l2);
                            const int r = A.rho[k];
                            if (t >= r) {
                                load_group(dst, lagp[k] + t + 1);
                            } else if (t + G3 <= r) {
                                load_group(dst, lagp[k] + t);
                            } else {
#pragma unroll
                                for (int u = 0; u < G3; ++u) dst[u] = lagp[k][t + u + ((t + u >= r) ? 1 : 0)];
                            }
                        }
                    };
                    auto body = [&](float (&o)[G3], float (&l0)[G3], float (&l1)[G3], float (&l2)[G3], int t) {
                        float ys[G3];
#pragma unroll
                        for (int u = 0; u < G3; ++u) {
                            y = trap_step<float, KIND>(y, o[u], l0[u], l1[u], l2[u], A.rr, A.ll);
                            ys[u] = y;
                        }
                        if ((capmask >> (t / G3)) & 1u) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int d = co[k] - t;
#pragma unroll
                                for (int u = 0; u < G3; ++u)
                                    if (d == u) capv[k] = ys[u];
                            }
                        }
                    };
                    fetch(a0, a1, a2, a3, 0);
#pragma unroll 1
                    for (int t = 0; t < C; t += 2 * G3) {
                        fetch(b0, b1, b2, b3, t + G3);
                        body(a0, a1, a2, a3, t);
                        if (t + 2 * G3 < C) fetch(a0, a1, a2, a3, t + 2 * G3);
                        body(b0, b1, b2, b3, t + G3);
                    }
                }
                // ---- true carries from the per-chunk increments (exact scan), then the pick-off
                const double D = (double)y - (double)g;
                const double delta = wave_exscan_add(D) - (double)g;
                float w4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = (float)((double)capv[k] + delta);
                    w4[k] = cl[k] >= 0 ? readlane(v, cl[k]) : 0.0f;
                }
                int fc = 0;
                result = pickoff_eval(t_in, A.mode, len, w4, fc);
                if (fc) report(fc, row);
            }
        }
        if (lane == 0) A.out[row * A.out_stride] = result;
        wave_sync();
    }
}

template <int KIND>
int launch_kind(const EnergyArgs& A, int npf, int64_t n_wf, int* err, int blocks, int threads, int lds_bytes, hipStream_t s) {
    switch (npf) {
        case 4: hipLaunchKernelGGL((dsp_energy_kernel<4, KIND>), dim3(blocks), dim3(threads), lds_bytes, s, A, n_wf, err); break;
        case 8: hipLaunchKernelGGL((dsp_energy_kernel<8, KIND>), dim3(blocks), dim3(threads), lds_bytes, s, A, n_wf, err); break;
        case 16: hipLaunchKernelGGL((dsp_energy_kernel<16, KIND>), dim3(blocks), dim3(threads), lds_bytes, s, A, n_wf, err); break;
        case 32: hipLaunchKernelGGL((dsp_energy_kernel<32, KIND>), dim3(blocks), dim3(threads), lds_bytes, s, A, n_wf, err); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

}  // namespace

// npf = number of 16-byte loads per lane that cover one waveform: 4, 8, 16 or 32 (C = 4*npf)
extern "C" int dsp_internal_launch_energy(const EnergyArgs* A, int trap_opcode, int npf, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, hipStream_t stream) {
    if (trap_opcode == DSP_OP_TRAP_FILTER) return launch_kind<TRAP_FILTER>(*A, npf, n_wf, err, blocks, threads, lds_bytes, stream);
    if (trap_opcode == DSP_OP_TRAP_NORM) return launch_kind<TRAP_NORM>(*A, npf, n_wf, err, blocks, threads, lds_bytes, stream);
    return launch_kind<TRAP_ASYM>(*A, npf, n_wf, err, blocks, threads, lds_bytes, stream);
}

extern "C" int dsp_internal_set_energy_lds(int trap_opcode, int npf, int lds_bytes) {
#define SET_(NPF, KIND)                                                                                                   \
    if (npf == NPF && kind == KIND)                                                                                        \
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_energy_kernel<NPF, KIND>),                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int kind = trap_opcode == DSP_OP_TRAP_FILTER ? TRAP_FILTER : (trap_opcode == DSP_OP_TRAP_NORM ? TRAP_NORM : TRAP_ASYM);
    SET_(4, TRAP_FILTER) SET_(8, TRAP_FILTER) SET_(16, TRAP_FILTER) SET_(32, TRAP_FILTER)
    SET_(4, TRAP_NORM) SET_(8, TRAP_NORM) SET_(16, TRAP_NORM) SET_(32, TRAP_NORM)
    SET_(4, TRAP_ASYM) SET_(8, TRAP_ASYM) SET_(16, TRAP_ASYM) SET_(32, TRAP_ASYM)
#undef SET_
    return (int)hipErrorInvalidValue;
}

extern "C" const char* dsp_internal_energy_kernel_name() { return "dsp_energy_kernel"; }
