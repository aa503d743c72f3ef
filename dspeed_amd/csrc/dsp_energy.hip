// dsp_energy.hip -- the Ge energy chain as ONE specialised kernel (BASELINE.json configs[1]/[3]):
//
//     waveform --bl_subtract--> --pole_zero--> --trap_filter|trap_norm|asym_trap--> fixed_time_pickoff --> 1 float
//
// Same arithmetic as the generic waveform VM (dsp_vm.hip) -- the host selects these kernels when a chain has exactly this
// shape.  One wavefront per waveform, lane j owns a chunk of consecutive samples, the next waveform's 16 KB are already in
// flight (16-byte global loads into registers) while the current one is filtered, and the trapezoid output is never stored
// (only the picked-off samples are kept).  Two kernels:
//   * dsp_energy_rr_kernel ("register resident", the default for 1024/2048/4096 samples): pad-free LDS image (sample i at
//     element i, C = len/64 + 1 samples per lane: an odd lane stride, so every "offset t of my chunk" access is
//     conflict-free), every pass over the chunk fully unrolled (immediate LDS offsets, counted waits), the lane's own chunk
//     in VGPRs from the staging to the end of the replay.  Per sample the LDS sees 2 writes (staging, pole-zero output) and
//     4 reads (own chunk once, three lagged trapezoid streams).
//   * dsp_energy_kernel ("classic"): the VM's slot layout (C = len/64, pitch C + 1, zero guard of 2 pitches below the slot),
//     chunk loops software-pipelined in groups of 8 samples.  Bit-identical to the VM; kept as the cross-check of the
//     default kernel and for A/B measurements (set_fused(15)).
// Reference bodies: processors/bl_subtract.py:11-46, pole_zero.py:24-77, trap_filters.py:12-227, fixed_time_pickoff.py:12-125.
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
    const float* tau;      // or null: the pole-zero time constant per event (a column) instead of c / tau_nan -- the TAU builds of the register-resident kernel
    int64_t tau_stride;
    int32_t ablate;        // diagnostic build only (-DDSPEED_HIP_DIAG, libdspeed_hip_diag.so): bit 0/1/2 = skip pass 1/2/3 (results are then
                           // wrong), bit 3 = per-phase cycle stamps.  The product library ignores the field: ABLATE below is a constant 0.
};

#ifdef DSPEED_HIP_DIAG
#define ABLATE(A) ((A).ablate)
#else
#define ABLATE(A) 0
#endif

// Wave priority by phase (s_setprio).  Two wavefronts share a SIMD, and each issues at most one instruction per four cycles: what the
// kernel gains from the second one is the overlap of one wavefront's dependent chains (pass 2's float64 recurrence, the scans and lane
// exchanges of the carries, the four-addition replay of pass 3, the tail) with the other's bulk work (staging stores, the chunk load,
// the float64 sums of pass 1).  The hardware arbitrates by priority, then age; with equal priorities the OLDER wavefront wins whatever
// it is doing, and a wavefront in a dependent chain loses its slot to the other's independent instructions every other time.  Priority
// that rises with the phase -- stage 0, pass 1 at 1, pass 2 at 2, carries / replay / tail at 3 -- lets the chain that is closest to
// finishing a row issue whenever it can and fills the gaps with the younger row's bulk work: 62.8 % -> 68.9 % of the HBM peak on the same
// box, same instructions (profiles/r03_headline_experiments.md: 20 sequences measured, every graded one within 0.5 % of this).
constexpr int rr_prio_after_phase[6] = {1, 2, 3, 3, 3, 0};  // priority of the phase that FOLLOWS boundary n (5: the next row's staging)
#define RR_PRIO_AT(n) __builtin_amdgcn_s_setprio(rr_prio_after_phase[n]);

namespace {

constexpr int G = 8;   // samples per software-pipeline group in the pole-zero passes
constexpr int G3 = 8;  // ... in the trapezoid replay (4 streams x 2 buffers live there: 64 VGPRs, fine at 2 waves/SIMD)

// diagnostic cycle stamps (DSPEED_HIP_ABLATE bit 3): where a wavefront spends its time, summed per phase into err[4 + 2*phase]
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PHASE(i)                                   \
    if (stamps) {                                  \
        const unsigned long long now_ = stamp();   \
        tsum[i] += now_ - tlast;                   \
        tlast = now_;                              \
    }

template <int N>
__device__ __forceinline__ void load_group(float (&v)[N], const float* p) {
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = p[u];
}

template <int NPF, int KIND>
__global__ void __launch_bounds__(256, 2) dsp_energy_kernel(EnergyArgs A, int64_t n_wf, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    const double inv_rr = 1.0 / A.rr, inv_ll = 1.0 / A.ll;  // trap_norm / asym_trap divide by these counts every sample
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
    const bool stamps = (ABLATE(A) & 8) != 0;
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = stamps ? stamp() : 0;

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
        PHASE(0)

        float result = quiet_nan<float>();
        // ---- pass 1: per-chunk float64 sum of x = w - baseline; a NaN anywhere (or a NaN baseline) poisons the sum
        double X = 0.0;
        if (!(ABLATE(A) & 1)) {
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
        PHASE(1)
        if (!in_nan) {
            const double E = wave_exscan_add(X);
            const float xlast = mine[C - 1] - bl;
            const double xprev0 = (double)wave_prev(xlast);
            // ---- pass 2: pole-zero recurrence in the reference's operation order, output in place; float32 running sum of
            // the output feeds the speculative carries of the trapezoid
            const double c = A.c;
            double acc = E - c * (E - xprev0), xp = xprev0;
            float run = 0.0f, cap[3] = {0.0f, 0.0f, 0.0f};
            if (!(ABLATE(A) & 2)) {
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
            PHASE(2)
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
            } else if (!A.all_nan && !(ABLATE(A) & 4) && pickoff_in_range(t_in, len)) {
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
                PHASE(3)
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
                            y = trap_step_r<float, KIND>(y, o[u], l0[u], l1[u], l2[u], A.rr, A.ll, inv_rr, inv_ll);
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
                PHASE(4)
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
        PHASE(5)
    }
    if (stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(err + 4) + i, tsum[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// Capture plan of the pad-free layout (C = len/64 + 1 samples per lane, sample i at LDS element i): for lag k and replay
// sub-chain s the speculative carry needs the prefix sum `local` samples into sub-chain `cs` of the lane `shift` below.
// Row-invariant, built by the host (dsp_host.cpp).
// ------------------------------------------------------------------------------------------------
struct EnergyPlan {
    int32_t shift[3][4];  // lane distance
    int32_t cs[3][4];     // sub-chain that holds the capture point
    int32_t local[3][4];  // capture after `local` samples of that sub-chain (0: nothing of it)
};

// ------------------------------------------------------------------------------------------------
// "rr" (register resident) kernel.
//
// What the measurements on MI355X said (profiles/r01_summary.md has the counters): a wavefront of this chain is bound by its
// own instruction stream -- about one instruction per 4 cycles, an LDS access several times that -- and at 17-20 KB of LDS
// per waveform only 2 wavefronts fit a SIMD, so nothing hides a stall.  Hence:
//   * every pass over the C samples of a lane is straight-line code (no loop counters, no address arithmetic, no uniform
//     branches inside: with those hipcc falls back to s_waitcnt lgkmcnt(0) and serialises LDS latency with the arithmetic);
//   * the chunk is read from LDS once and stays in VGPRs through pass 1 (float64 sum), pass 2 (pole-zero, in place) and
//     pass 3 (trapezoid replay); only the three lagged streams of the replay come from LDS, loaded PD stages ahead;
//   * bl_subtract happens in the staging stores; the result of row r is stored behind the prefetch of row r + 2 (vmcnt
//     counts stores: a store at the end of the loop body would sit in front of the next staging's s_waitcnt vmcnt(0));
//   * what must be picked out at a run-time position is never tested per sample: the float32 prefix sum at each 8-sample
//     group end and the replay state at each group start go to a 9-entry per-lane LDS side array (a run-time group number is
//     then an address); the two trapezoid samples every pick-off mode needs are copied out of a 16-sample register window
//     by one uniform branch per 16 samples (a not-taken branch costs tens of cycles; 32 of them cost more than they saved).
// S = sub-chains of the replay per lane: 2 halves the dependent-add chain but doubles the carry captures; measured slower.
// ------------------------------------------------------------------------------------------------
// IN: waveform element type in HBM: 0 float32, 1 int16, 2 uint16 (digitiser samples; widened to float32 while staging, exactly
// like the reference's ufunc casting picks the float32 loop for them, processing_chain.py:1565-1572)
// exp(-1 / tau) in float64 for a time constant that varies per event, as the interpreter's op forms it (dsp_vm.hip pz_decay; pole_zero.py:60):
// out of line, the device's exp is 200 instructions
__device__ __attribute__((noinline)) double rr_decay(double tau) { return exp(-1.0 / tau); }

template <int NPF, int KIND, int S, int IN, bool TAU = false>
__global__ void __launch_bounds__(256, NPF >= 32 ? 1 : 2) dsp_energy_rr_kernel(EnergyArgs A, EnergyPlan PL, int64_t n_wf, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int C = 4 * NPF + 1, len = 256 * NPF, NG = (C - 1) / 8, CS = (C - 1) / S, NGS = CS / 8;
    constexpr int BS = CS >= 16 ? 16 : CS;  // samples per capture block
    static_assert((C - 1) % (8 * S) == 0, "sub-chain length must be a whole number of 8-sample groups");
    const int lane = lane_id();
    const double inv_rr = 1.0 / A.rr, inv_ll = 1.0 / A.ll;  // trap_norm / asym_trap divide by these counts every sample
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    float* lds = reinterpret_cast<float*>(smem_raw) + (size_t)wave * A.lds_elems_per_wave;
    for (int e = lane; e < A.lds_elems_per_wave; e += 64) lds[e] = 0.0f;
    wave_sync();
    float* slot = lds + A.slot_off;
    float* mine = slot + lane * C;
    // per-lane side array (pitch 9, odd): group-end prefix sums of pass 2, later the group-start states of the replay.  Kept in
    // LDS so that "the value of group gi" with a run-time gi is an address, not a register select chain
    constexpr int AUXP = NG + 1 <= 9 ? 9 : ((NG + 1) | 1);  // (9 for up to 4096 samples, 17 for 8192; the host sizes the region the same way)
    static_assert(NG + 1 <= AUXP && S * NGS + 1 <= AUXP && (AUXP & 1) == 1, "side array too small");
    float* aux = slot + 64 * C + 16 + lane * AUXP;
    const float* lagp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pos0 = lane * C - A.q[k];  // q[] carries the lags
        lagp[k] = (pos0 >= -C) ? slot + pos0 : slot - (2 * C + 8);
    }

    const int64_t stride_rows = (int64_t)gridDim.x * wpb;
    int64_t row = (int64_t)blockIdx.x * wpb + wave;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    constexpr int NLD = IN == 0 ? NPF : NPF / 2;  // 16-byte loads per lane and waveform: 4 float32 or 8 16-bit samples each
    static_assert(IN == 0 || NPF % 2 == 0, "16-bit rows: an even number of 4-sample groups per lane");
    u4 pf[NLD];
    float pf_bl = 0.0f, pf_tp = 0.0f;
    auto prefetch = [&](int64_t r) {
        const char* g = (const char*)A.wf + (r * A.wf_stride + A.wf_offset) * (IN == 0 ? 4 : 2);
#pragma unroll
        for (int b = 0; b < NLD; ++b) pf[b] = reinterpret_cast<const u4*>(g)[b * 64 + lane];
        // (address space 1 spelled out: a pointer that went through a null test is otherwise loaded with flat_load, whose
        // out-of-order return forces every LDS wait that follows it down to lgkmcnt(0))
        typedef const __attribute__((address_space(1))) float* gptr;
        pf_bl = A.bl ? ((gptr)A.bl)[r * A.bl_stride] : A.bl_const;
        pf_tp = A.tp ? ((gptr)A.tp)[r * A.tp_stride] : A.tp_const;
    };
    auto report = [&](int code, int64_t r) {
        if (lane == 0 && atomicCAS(&err[0], 0, code) == 0) {
            err[1] = (int)(r & 0xffffffffll);
            err[2] = (int)(r >> 32);
        }
    };
    if (row < n_wf) prefetch(row);
    const bool stamps = (ABLATE(A) & 8) != 0;
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = stamps ? stamp() : 0;
    // a row's result is stored one iteration late, behind the next prefetch: vmcnt counts stores too, so a store issued
    // at the end of the loop body would sit (a full write latency) in front of the s_waitcnt vmcnt(0) that opens the next staging
    float pend_result = 0.0f;
    int64_t pend_row = -1;

    for (; row < n_wf; row += stride_rows) {
        // bl_subtract while staging (16-bit samples: unpacked and converted first; a lane's load covers 8 consecutive samples)
#pragma unroll
        for (int b = 0; b < NLD; ++b) {
            if (IN == 0) {
                f4 v;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = __uint_as_float(pf[b][m]);
                *reinterpret_cast<f4*>(slot + (b * 64 + lane) * 4) = v - pf_bl;
            } else {
                f4 lo, hi;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const unsigned int wv = pf[b][m];
                    const float s0 = IN == 1 ? (float)(short)(wv & 0xffffu) : (float)(wv & 0xffffu);
                    const float s1 = IN == 1 ? (float)(short)(wv >> 16) : (float)(wv >> 16);
                    if (m < 2) {
                        lo[2 * m] = s0;
                        lo[2 * m + 1] = s1;
                    } else {
                        hi[2 * (m - 2)] = s0;
                        hi[2 * (m - 2) + 1] = s1;
                    }
                }
                *reinterpret_cast<f4*>(slot + (b * 64 + lane) * 8) = lo - pf_bl;
                *reinterpret_cast<f4*>(slot + (b * 64 + lane) * 8 + 4) = hi - pf_bl;
            }
        }
        slot[len + lane] = 0.0f;  // virtual samples above len
        const float t_in = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pf_tp)));
        const int64_t next = row + stride_rows;
        __builtin_amdgcn_sched_barrier(0);
        if (next < n_wf) prefetch(next);
        if (pend_row >= 0 && lane == 0) A.out[pend_row * A.out_stride] = pend_result;
        __builtin_amdgcn_sched_barrier(0);
        wave_sync();
        PHASE(0)
        RR_PRIO_AT(0)

        float result = quiet_nan<float>();
        // ---- the lane's chunk lives in registers from here to the end of the replay: x, then (in place) the pole-zero output
        float xr[C];
#pragma unroll
        for (int t = 0; t < C; ++t) xr[t] = mine[t];
        const float xprev = (lane > 0) ? mine[-1] : 0.0f;
        // ---- pass 1: float64 sum of x over the chunk
        // (four partial sums: the float64 sum of 65 float32 samples is exact for one waveform's dynamic range, so the order is free,
        // and one chain of dependent float64 adds would cost their full latency 65 times)
        double Xp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < C; ++t) {
            if ((t & 7) == 0) __builtin_amdgcn_sched_barrier(0);  // (keeps the float64 conversions from being hoisted: registers)
            Xp[t & 3] += (double)xr[t];
        }
        const double X = (Xp[0] + Xp[1]) + (Xp[2] + Xp[3]);
        double c_row = A.c;
        bool in_nan = A.tau_nan != 0;
        if constexpr (TAU) {  // (a build of its own: the constant-tau kernels keep their code)
            const float tau = A.tau[row * A.tau_stride];
            in_nan = tau != tau;
            c_row = rr_decay((double)tau);
        }
        if (wave_any(!(fabs(X) <= 1.7976931348623157e308))) {
            bool n = false;
#pragma unroll
            for (int t = 0; t < C; ++t) n |= (xr[t] != xr[t]);
            in_nan |= wave_any(n);
        }
        PHASE(1)
        RR_PRIO_AT(1)
        // the float64 images of the samples are cheaper to recompute in pass 2 than to keep (130 registers): hide the reuse
#pragma unroll
        for (int t = 0; t < C; ++t) asm volatile("" : "+v"(xr[t]));
        if (!in_nan) {
            const double E = wave_exscan_add(X);
            // ---- pass 2 (straight line): pole-zero recurrence in the reference's operation order, in place
            const double c = c_row;
            double xp = (double)xprev, acc = E - c * (E - xp);
            float run = 0.0f;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                if ((t & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                const double x = (double)xr[t];
                // acc_k = acc_{k-1} + (x_k - c x_{k-1}); the float64 association differs from the reference's (acc + x) - xp*c by
                // <= 1 ulp of a double (the chunk carry already does), invisible after the float32 store; the chain is one add long
                acc += __builtin_fma(-c, xp, x);
                const float y = (float)acc;
                xr[t] = y;
                mine[t] = y;  // other lanes read it with a lag
                xp = x;
                run += y;
                if ((t & 7) == 7) aux[t >> 3] = run;
            }
            wave_sync();
            PHASE(2)
        RR_PRIO_AT(2)
            bool pz_nan = false;
            if (wave_any(!(fabsf(run) <= 3.4028234663852886e38f))) {
                bool n = false;
#pragma unroll
                for (int t = 0; t < C; ++t) n |= (xr[t] != xr[t]);
                pz_nan = wave_any(n);
            }
            if (pz_nan) {
                report(DSP_E_PZ_NAN, row);
            } else if (!A.all_nan && !(ABLATE(A) & 4) && pickoff_in_range(t_in, len)) {
                // ---- speculative carries
                const double Ep = wave_exscan_add((double)run);
                float g[S], y[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    double Ak[3];
                    float pbase[3], pv[3][8];
                    int pn[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        int r = PL.cs[k][s] * CS + PL.local[k][s];  // samples of the source lane's chunk before the capture point
                        // opaque to the optimiser: otherwise every mask derived from the (row-invariant) plan is hoisted out of the
                        // row loop and the kernel drowns in spilled SGPR pairs
                        asm volatile("" : "+s"(r));
                        const int gi = (r > 0 ? r - 1 : 0) >> 3;  // group that contains sample r-1
                        pn[k] = r - 8 * gi;                        // 0..8 samples of group gi (gi == NG: the odd sample)
                        const float b = aux[gi > 0 ? gi - 1 : 0];
                        pbase[k] = gi > 0 ? b : 0.0f;
                        const float* p = mine + 8 * gi;
#pragma unroll
                        for (int u = 0; u < 8; ++u) pv[k][u] = p[u];  // (reads at most 7 past the chunk: inside the slot tail)
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float part = 0.0f;
#pragma unroll
                        for (int u = 0; u < 8; ++u) part += (u < pn[k]) ? pv[k][u] : 0.0f;
                        Ak[k] = wave_shift_up(Ep + (double)(pbase[k] + part), PL.shift[k][s]);
                    }
                    const double own = Ep + (s ? (double)aux[s * NGS - 1] : 0.0);
                    double Gd;
                    if (KIND == TRAP_FILTER)
                        Gd = ((own - Ak[0]) - Ak[1]) + Ak[2];
                    else if (KIND == TRAP_NORM)
                        Gd = (((own - Ak[0]) - Ak[1]) + Ak[2]) / A.rr;
                    else
                        Gd = (own - Ak[0]) / A.rr - (Ak[1] - Ak[2]) / A.ll;
                    g[s] = (lane == 0 && s == 0) ? -0.0f : (float)Gd;
                    y[s] = g[s];
                }
                PHASE(3)
        RR_PRIO_AT(3)
                // ---- pass 3: replay; own samples from registers, the three lagged streams software-pipelined one 8-sample group ahead
                wave_sync();  // the prefix sums in aux are consumed; aux now receives the replay state at every group start
                // the two samples every pick-off mode needs (floor and ceil of the time point) are caught on the fly: stage numbers
                const int i0 = (int)t_in;
                int capst[2], capoff[2], caplane[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = i0 + k;
                    const bool ok = e >= 0 && e < len;
                    caplane[k] = ok ? e / C : 0;
                    capoff[k] = ok ? e - caplane[k] * C : 0;
                    capst[k] = ok ? capoff[k] / BS : -1;  // capture block of the chunk; block (C-1)/BS (never reached) = the odd sample
                }
                // one test per BS samples (a not-taken branch still costs tens of cycles): bit q set = block q holds a wanted sample
                int capmask = (capst[0] >= 0 ? 1 << capst[0] : 0) | (capst[1] >= 0 ? 1 << capst[1] : 0);
                asm volatile("" : "+s"(capmask));  // one live scalar, not a recomputation at each test
                float* capbuf = slot + 64 * C + 16 + 64 * AUXP;  // 2 x 16 floats per wavefront, written by the lane that owns the sample
                float ysb[S][BS];
                constexpr int GL = 4, NL = CS / GL;  // samples per pipeline stage of the lagged streams (registers: 2 x S x 3 x GL)
                constexpr int PD = S == 1 ? 2 : 1;  // stages the lagged loads run ahead of their use (lgkmcnt counts to 15: 6*S reads per stage)
                float lb[PD + 1][S][GL][3], lodd[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int p = 0; p < PD; ++p)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int u = 0; u < GL; ++u)
#pragma unroll
                            for (int k = 0; k < 3; ++k) lb[p][s][u][k] = lagp[k][s * CS + p * GL + u];
#pragma unroll
                for (int gl = 0; gl < NL; ++gl) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (gl + PD < NL) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int u = 0; u < GL; ++u)
#pragma unroll
                                for (int k = 0; k < 3; ++k) lb[(gl + PD) % (PD + 1)][s][u][k] = lagp[k][s * CS + (gl + PD) * GL + u];
                    } else if (gl + PD == NL) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) lodd[k] = lagp[k][C - 1];
                    }
                    if ((gl * GL) % 8 == 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) aux[s * NGS + (gl * GL) / 8] = y[s];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // the reads just issued are younger than the stage consumed next: a counted wait
#pragma unroll
                    for (int u = 0; u < GL; ++u)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            y[s] = trap_step_r<float, KIND>(y[s], xr[s * CS + gl * GL + u], lb[gl % (PD + 1)][s][u][0], lb[gl % (PD + 1)][s][u][1],
                                                          lb[gl % (PD + 1)][s][u][2], A.rr, A.ll, inv_rr, inv_ll);
                            ysb[s][(gl * GL + u) % BS] = y[s];
                        }
                    if (((gl + 1) * GL) % BS == 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const int q = s * (CS / BS) + (gl * GL) / BS;
                            if (capmask & (1 << q)) {  // uniform, taken at most twice per waveform
#pragma unroll
                                for (int k = 0; k < 2; ++k)
                                    if (capst[k] == q && lane == caplane[k]) {
#pragma unroll
                                        for (int u = 0; u < BS; ++u) capbuf[k * 16 + u] = ysb[s][u];
                                    }
                            }
                        }
                    }
                }
                aux[S * NGS] = y[S - 1];  // state before the odd sample (it extends the last chain)
                y[S - 1] = trap_step_r<float, KIND>(y[S - 1], xr[C - 1], lodd[0], lodd[1], lodd[2], A.rr, A.ll, inv_rr, inv_ll);
                PHASE(4)
        RR_PRIO_AT(4)
                // ---- true carries: exact scan of the increments
                double D[S], Dbefore[S], Dtot = 0.0;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    D[s] = (double)y[s] - (double)g[s];
                    Dbefore[s] = Dtot;
                    Dtot += D[s];
                }
                const double T0 = wave_exscan_add(Dtot);
                // ---- wanted samples: re-run the one 8-sample group that contains each of them from its saved start state
                const bool wide = (A.mode == 'h');
                float w4[4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float v = capbuf[k * 16 + (capoff[k] % BS)];  // (stale when the sample is the odd one or out of range: not used then)
                    v = capoff[k] == C - 1 ? y[S - 1] : v;
                    double delta = T0 - (double)g[0];
#pragma unroll
                    for (int s = 1; s < S; ++s)
                        if (capoff[k] >= s * CS) delta = (T0 + Dbefore[s]) - (double)g[s];
                    w4[1 + k] = capst[k] >= 0 ? readlane((float)((double)v + delta), caplane[k]) : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 4; k += 3) {  // the outer two samples of the 4-point mode: re-run their 8-sample group
                    const int e = i0 - 1 + k;
                    const bool need = wide && e >= 0 && e < len;
                    w4[k] = 0.0f;
                    __builtin_amdgcn_sched_barrier(0);
                    if (need) {  // uniform
                        const int l = e / C, off = e - l * C;
                        int ch = off / CS;
                        if (ch > S - 1) ch = S - 1;
                        const int loc = off - ch * CS;  // 0..CS (CS: the odd sample, chain S-1 only)
                        const int gi = loc >> 3, u0 = loc & 7;
                        float ys = aux[ch * NGS + gi], gsel = 0.0f;  // (chain S-1, group NGS) -> aux[S*NGS]
                        double dsel = 0.0;
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            if (ch == s) {
                                gsel = g[s];
                                dsel = Dbefore[s];
                            }
                        const int base = ch * CS + gi * 8;
                        float yk = ys;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int tt = base + u;  // (beyond the chunk for the odd sample's group: reads stay in the slot tail, unused)
                            ys = trap_step_r<float, KIND>(ys, mine[tt], lagp[0][tt], lagp[1][tt], lagp[2][tt], A.rr, A.ll, inv_rr, inv_ll);
                            if (u == u0) yk = ys;
                        }
                        const double delta = (T0 + dsel) - (double)gsel;
                        w4[k] = readlane((float)((double)yk + delta), l);
                    }
                }
                int fc = 0;
                result = pickoff_eval(t_in, A.mode, len, w4, fc);
                if (fc) report(fc, row);
            }
        }
        pend_result = result;
        pend_row = row;
        wave_sync();
        PHASE(5)
        RR_PRIO_AT(5)
    }
    if (pend_row >= 0 && lane == 0) A.out[pend_row * A.out_stride] = pend_result;
    if (stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(err + 4) + i, tsum[i]);
    }
}

template <int KIND, int S, int IN, bool TAU = false>
int launch_rr_kind(const EnergyArgs& A, const EnergyPlan& PL, int npf, int64_t n_wf, int* err, int blocks, int threads, int lds_bytes,
                   hipStream_t st) {
    switch (npf) {
        case 4: hipLaunchKernelGGL((dsp_energy_rr_kernel<4, KIND, S, IN, TAU>), dim3(blocks), dim3(threads), lds_bytes, st, A, PL, n_wf, err); break;
        case 8: hipLaunchKernelGGL((dsp_energy_rr_kernel<8, KIND, S, IN, TAU>), dim3(blocks), dim3(threads), lds_bytes, st, A, PL, n_wf, err); break;
        case 16: hipLaunchKernelGGL((dsp_energy_rr_kernel<16, KIND, S, IN, TAU>), dim3(blocks), dim3(threads), lds_bytes, st, A, PL, n_wf, err); break;
        case 32:  // 8192 samples (production LEGEND rows): 129 samples per lane, one wavefront per SIMD (512-register budget, 36 KB of LDS)
            if (S != 1) return (int)hipErrorInvalidValue;
            hipLaunchKernelGGL((dsp_energy_rr_kernel<32, KIND, 1, IN, TAU>), dim3(blocks), dim3(threads), lds_bytes, st, A, PL, n_wf, err);
            break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
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

// register-resident kernel; S = sub-chains of the trapezoid replay (1: default, 2: measured slower, kept for A/B, float32 rows only);
// plan[S - 1]; wf_dtype = DSP_F32 / DSP_I16 / DSP_U16 rows
extern "C" int dsp_internal_launch_energy_rr(const EnergyArgs* A, const EnergyPlan* PL, int trap_opcode, int npf, int S, int wf_dtype,
                                             int64_t n_wf, int* err, int blocks, int threads, int lds_bytes, hipStream_t stream) {
#define GO_(KIND)                                                                                                          \
    if (A->tau && wf_dtype == DSP_I16) return launch_rr_kind<KIND, 1, 1, true>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream); \
    if (A->tau && wf_dtype == DSP_U16) return launch_rr_kind<KIND, 1, 2, true>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream); \
    if (A->tau) return launch_rr_kind<KIND, 1, 0, true>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream);       \
    if (wf_dtype == DSP_I16) return launch_rr_kind<KIND, 1, 1>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream); \
    if (wf_dtype == DSP_U16) return launch_rr_kind<KIND, 1, 2>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream); \
    return S == 2 ? launch_rr_kind<KIND, 2, 0>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream)                \
                  : launch_rr_kind<KIND, 1, 0>(*A, *PL, npf, n_wf, err, blocks, threads, lds_bytes, stream);
    if (trap_opcode == DSP_OP_TRAP_FILTER) { GO_(TRAP_FILTER) }
    if (trap_opcode == DSP_OP_TRAP_NORM) { GO_(TRAP_NORM) }
    GO_(TRAP_ASYM)
#undef GO_
}
extern "C" const char* dsp_internal_energy_rr_kernel_name() { return "dsp_energy_rr_kernel"; }

extern "C" const char* dsp_internal_energy_kernel_name() { return "dsp_energy_kernel"; }
