/* dsp_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see dsp_oracle.h for the rules of use).
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include "dsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T float
#define SFX f32
#include "dsp_oracle_impl.h"
#undef T
#undef SFX

#define T double
#define SFX f64
#include "dsp_oracle_impl.h"
#undef T
#undef SFX

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* One block of the energy chain: the four gufunc calls ProcessorManager.execute would make on a
 * (block_width, len) buffer set (processing_chain.py:1778-1781), in recipe order. */
static int energy_block(const float* wf, long rows, int len, const float* bl, const float* tp, float tau, int rise, int flat,
                        int mode, float* e_out, float* s_bl, float* s_pz, float* s_tr) {
    int rc = 0, e;
    long bad;
    if ((e = orc_bl_subtract_f32(wf, rows, len, bl, 1, s_bl, &bad)) && !rc) rc = e;
    if ((e = orc_pole_zero_f32(s_bl, rows, len, tau, s_pz, &bad)) && !rc) rc = e;
    if ((e = orc_trap_filter_f32(s_pz, rows, len, rise, flat, s_tr, &bad)) && !rc) rc = e;
    if ((e = orc_fixed_time_pickoff_f32(s_tr, rows, len, tp, 1, mode, e_out, &bad)) && !rc) rc = e;
    return rc;
}

int orc_chain_energy_f32(const float* wf, long n_wf, int len, const float* baseline, const float* t_pick, float tau, int rise,
                         int flat, int mode, float* e_out, int block_width, int n_threads) {
    if (block_width <= 0) block_width = 16;
    long n_blocks = (n_wf + block_width - 1) / block_width;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 1 ? n_threads : 1)
#endif
    {
        size_t sz = (size_t)block_width * (size_t)len;
        float* s_bl = (float*)malloc(sizeof(float) * sz);
        float* s_pz = (float*)malloc(sizeof(float) * sz);
        float* s_tr = (float*)malloc(sizeof(float) * sz);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long b = 0; b < n_blocks; ++b) {
            long r0 = b * block_width, rows = (n_wf - r0 < block_width) ? n_wf - r0 : block_width;
            int e = energy_block(wf + r0 * len, rows, len, baseline + r0, t_pick + r0, tau, rise, flat, mode, e_out + r0, s_bl,
                                 s_pz, s_tr);
            if (e) {
#ifdef _OPENMP
#pragma omp critical
#endif
                if (!rc) rc = e;
            }
        }
        free(s_bl);
        free(s_pz);
        free(s_tr);
    }
    return rc;
}

int orc_chain_pz_trap_f32(const float* wf, long n_wf, int len, float tau, int rise, int flat, float* out, int block_width,
                          int n_threads) {
    if (block_width <= 0) block_width = 16;
    long n_blocks = (n_wf + block_width - 1) / block_width;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 1 ? n_threads : 1)
#endif
    {
        float* s_pz = (float*)malloc(sizeof(float) * (size_t)block_width * (size_t)len);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long b = 0; b < n_blocks; ++b) {
            long r0 = b * block_width, rows = (n_wf - r0 < block_width) ? n_wf - r0 : block_width, bad;
            int e = orc_pole_zero_f32(wf + r0 * len, rows, len, tau, s_pz, &bad);
            int e2 = orc_trap_filter_f32(s_pz, rows, len, rise, flat, out + r0 * len, &bad);
            if (e || e2) {
#ifdef _OPENMP
#pragma omp critical
#endif
                if (!rc) rc = e ? e : e2;
            }
        }
        free(s_pz);
    }
    return rc;
}
