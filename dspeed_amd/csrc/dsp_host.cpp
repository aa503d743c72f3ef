// dsp_host.cpp -- host half of libdspeed_hip.so: the C ABI of include/dspeed_hip.h.
//
//   * device/memory/stream helpers (thin wrappers so that a host needs nothing but this library);
//   * chain translation: validates a dsp_op program, evaluates every constant the reference evaluates once per
//     call in float64 on the host (exp(-1/tau) through libm exactly as numba/LLVM does, IIR coefficients, lag
//     splits), lays the waveform slots out in LDS and picks the launch geometry;
//   * the single-processor entry points (dsp_<name>_f32), which are tiny cached chains.
//
// Reference behaviour mirrored here: constant-only DSPFatal conditions are raised at chain creation with the
// reference's own codes/messages (processors/*.py, cited in include/dspeed_hip.h).
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "dsp_plan.h"

extern "C" int dsp_internal_launch_vm_f32(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, int with_fir, int team, hipStream_t stream);
extern "C" int dsp_internal_launch_vm_f64(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, int with_fir, hipStream_t stream);
extern "C" int dsp_internal_set_vm_lds(int lds_bytes);
extern "C" const char* dsp_internal_vm_kernel_name();
extern "C" int dsp_internal_launch_stream_read(const void* src, int64_t bytes, uint32_t* sink, int blocks, hipStream_t stream);
extern "C" int dsp_internal_launch_fit_rows(const FitArgs* A, int wf_dtype, int compute_dtype, hipStream_t stream);
extern "C" int dsp_internal_launch_synth(void* wf, int out_dtype, int64_t n_wf, int wf_len, int64_t row_stride, float* baseline,
                                         float* t_pick, uint64_t seed, int64_t first_row, float tau, float sigma, float pick_offset,
                                         float bl_lo, float bl_hi, float amp_lo, float amp_hi, float rise_lo, float rise_hi, hipStream_t stream);

extern "C" int dsp_internal_launch_energy(const EnergyArgs* A, int trap_opcode, int npf, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_set_energy_lds(int trap_opcode, int npf, int lds_bytes);
extern "C" const char* dsp_internal_energy_kernel_name();
extern "C" int dsp_internal_launch_energy_rr(const EnergyArgs* A, const EnergyPlan* PL, int trap_opcode, int npf, int S, int wf_dtype,
                                             int64_t n_wf, int* err, int blocks, int threads, int lds_bytes, hipStream_t stream);
extern "C" const char* dsp_internal_energy_rr_kernel_name();
extern "C" int dsp_internal_launch_rows(const RowsArgs* A, int64_t n_wf, int* err, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_set_rows_lds(int lds_bytes);
extern "C" const char* dsp_internal_rows_kernel_name();
extern "C" int dsp_internal_launch_pz_rows(const PzArgs* A, int64_t n_wf, int* err, hipStream_t stream);
extern "C" const char* dsp_internal_pz_rows_kernel_name();
extern "C" int dsp_internal_launch_reduce(const ReduceArgs* A, int64_t n_wf, int dtype, int vec, int* err, hipStream_t stream);
extern "C" const char* dsp_internal_reduce_kernel_name();
extern "C" int dsp_internal_launch_scalar(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int n_sregs, int type, hipStream_t stream);  // type: 0 float32, 1 float64, 2 int64 registers
extern "C" int dsp_internal_set_scalar_lds(int lds_bytes);
extern "C" const char* dsp_internal_scalar_kernel_name();
extern "C" int dsp_internal_current_lds_bytes(int ma_len);
extern "C" int dsp_internal_launch_fir_runs(const FirRunsArgs* A, FirRunsTable* table, int64_t n_wf, int blocks, int* err, hipStream_t stream);
extern "C" int dsp_internal_launch_current(const CurrentArgs* A, int64_t n_wf, int blocks, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_set_current_lds(int lds_bytes);
extern "C" const char* dsp_internal_current_kernel_name();
extern "C" int dsp_internal_fir_mfma_lds_bytes(int kend);
extern "C" int dsp_internal_launch_fir_mfma(const FirArgs* A, int64_t n_wf, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_launch_fir_f16(const FirArgs* A, const FirF16Taps* T, int64_t n_wf, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_fir_f16_tz(int kend);
extern "C" size_t dsp_internal_fir_f16_taps_bytes(int kend);
extern "C" int dsp_internal_fir_f16_lds_bytes();
extern "C" int dsp_internal_set_fir_f16_lds(int lds_bytes);
extern "C" const char* dsp_internal_fir_f16_kernel_name();
extern "C" int dsp_internal_set_fir_mfma_lds(int lds_bytes);
extern "C" const char* dsp_internal_fir_mfma_kernel_name();
extern "C" int dsp_internal_fir_store_lds_bytes(int kend);
extern "C" int dsp_internal_launch_fir_store(const FirArgs* A, int64_t n_wf, int lds_bytes, hipStream_t stream);
extern "C" int dsp_internal_set_fir_store_lds(int lds_bytes);
extern "C" const char* dsp_internal_fir_store_kernel_name();

#define fail dsp_fail
#define elem_size dsp_elem_size

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError(); /* reported here: do not leave it for an unrelated later launch check */ \
            return fail(DSP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));               \
        }                                                                                          \
    } while (0)

// Release-type calls (free, destroy, unregister) run from destructors / garbage collection at arbitrary moments of the caller: they
// report through their return code only and leave dsp_last_error() alone, so the text of an error the caller is about to read is
// never replaced by an unrelated one.
#define HIP_RELEASE(expr)                       \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) {                 \
            (void)hipGetLastError();            \
            return DSP_ERR_HIP;                 \
        }                                       \
    } while (0)

struct dsp_chain : ChainPlan {
    DevProgram* dev = nullptr;
    size_t dev_bytes = 0;      // bytes of the device copy (the used prefix of DevProgram)
    int* dev_err = nullptr;
    int device = 0;
    int num_cu = 256;
    int64_t f16_rows_cap = 0;
    // dsp_chain_share_row_scales: the pole-zero rows kernel in front writes the scales and flags of the rows it stores straight into this
    // chain's arrays and leaves a note of which rows they describe; dsp_chain_execute checks the note against its own input
    dsp_chain* scale_feeder = nullptr;  // (on the float16 FIR chain)
    dsp_chain* scale_sink = nullptr;    // (on the pole-zero rows chain)
    // the same for a float16 FIR that reads a slice of the rows the pole-zero kernel READS, minus the same baseline column: the kernel sees
    // x - baseline of every sample anyway
    dsp_chain* scale_sink_in = nullptr;  // (on the pole-zero rows chain)
    bool fed_in_side = false;            // (on the FIR chain: the note below describes its rows as the producer's input, fed_bl its baseline)
    const void* fed_bl = nullptr;
    const void* fed_rows_ptr = nullptr;
    void* fed_stream = nullptr;  // the stream of the producer's launch: its scales and flags are ordered ahead of a consumer on that stream only
    int64_t fed_n_wf = -1, fed_stride = 0;
    int32_t fed_len = 0;
    float* cur_scratch = nullptr;  // (current-branch kernel) allocated at the first launch
    int cur_blocks_cap = 0;
    FirRunsTable* runs_table = nullptr;  // (run-length FIR) the kernel's breakpoints, rewritten by every launch; allocated at the first
    float* runs_scratch = nullptr;       // a row per resident wavefront for a filtered waveform that nothing keeps
    // the error word handed to the host by a copy that is part of the launch (dsp_chain_set_async_check): dsp_chain_check then needs no
    // transfer of its own -- one issued while a large host-to-device copy of the next buffer is in flight queues up behind it
    int* err_mirror = nullptr;  // page-locked
    ~dsp_chain() {  // (also on the error paths of dsp_chain_create, which holds the chain in a unique_ptr)
        if (scale_sink) scale_sink->scale_feeder = nullptr;
        if (scale_sink_in) scale_sink_in->scale_feeder = nullptr;
        if (scale_feeder) {
            if (scale_feeder->scale_sink == this) scale_feeder->scale_sink = nullptr;
            if (scale_feeder->scale_sink_in == this) scale_feeder->scale_sink_in = nullptr;
        }
        if (dev) (void)hipFree(dev);
        if (dev_err) (void)hipFree(dev_err);
        if (host.prof) (void)hipFree(host.prof);
        if (err_mirror) (void)hipHostFree(err_mirror);
        if (cur_scratch) (void)hipFree(cur_scratch);
        if (runs_table) (void)hipFree(runs_table);
        if (runs_scratch) (void)hipFree(runs_scratch);
        for (int k = 0; k < DSP_FIR_MAXK; ++k)
            if (f16.taps16[k]) (void)hipFree(const_cast<void*>(f16.taps16[k]));
        if (f16.row_scale) (void)hipFree(const_cast<void*>(f16.row_scale));
        if (f16.row_flags) (void)hipFree(const_cast<void*>(f16.row_flags));
    }
};

// Internal copies between this library's own host structures (programs, error words, tap read-backs) and the device go through one
// page-locked staging buffer: the runtime is never handed heap or stack memory to page-lock on the fly (host memory note).
static std::mutex g_stage_mu;
static void* g_stage = nullptr;
static const size_t STAGE_BYTES = 1 << 20;
static hipError_t stage_ready() {
    if (g_stage) return hipSuccess;
    return hipHostMalloc(&g_stage, STAGE_BYTES, hipHostMallocDefault);
}
static hipError_t staged_h2d(void* dev, const void* host, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    hipError_t e = stage_ready();
    for (size_t o = 0; e == hipSuccess && o < bytes; o += STAGE_BYTES) {
        const size_t n = bytes - o < STAGE_BYTES ? bytes - o : STAGE_BYTES;
        memcpy(g_stage, (const char*)host + o, n);
        e = hipMemcpy((char*)dev + o, g_stage, n, hipMemcpyHostToDevice);
    }
    return e;
}
static hipError_t staged_d2h(void* host, const void* dev, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    hipError_t e = stage_ready();
    for (size_t o = 0; e == hipSuccess && o < bytes; o += STAGE_BYTES) {
        const size_t n = bytes - o < STAGE_BYTES ? bytes - o : STAGE_BYTES;
        e = hipMemcpy(g_stage, (const char*)dev + o, n, hipMemcpyDeviceToHost);
        if (e == hipSuccess) memcpy((char*)host + o, g_stage, n);
    }
    return e;
}

extern "C" {

// ------------------------------------------------------------------------------------------------ device / memory
int dsp_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(DSP_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return DSP_OK;
}
int dsp_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return DSP_OK;
}
int dsp_get_device(int* device) {
    HIP_TRY(hipGetDevice(device));
    return DSP_OK;
}
int dsp_device_info(int device, char* name, int name_cap, int* compute_units, int64_t* hbm_bytes, int* lds_bytes_per_cu) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    return DSP_OK;
}
int dsp_malloc(void** dev, int64_t bytes) {
    *dev = nullptr;
    HIP_TRY(hipMalloc(dev, (size_t)(bytes > 0 ? bytes : 1)));
    return DSP_OK;
}
int dsp_free(void* dev) {
    if (dev) HIP_RELEASE(hipFree(dev));
    return DSP_OK;
}
int dsp_host_alloc(void** host, int64_t bytes) {
    *host = nullptr;
    HIP_TRY(hipHostMalloc(host, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault));
    return DSP_OK;
}
int dsp_host_free(void* host) {
    if (host) HIP_RELEASE(hipHostFree(host));
    return DSP_OK;
}
int dsp_memset(void* dev, int value, int64_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(dev, value, (size_t)bytes, (hipStream_t)stream));
    return DSP_OK;
}
int dsp_h2d(void* dev, const void* host, int64_t bytes) {
    HIP_TRY(hipMemcpy(dev, host, (size_t)bytes, hipMemcpyHostToDevice));
    return DSP_OK;
}
int dsp_d2h(void* host, const void* dev, int64_t bytes) {
    HIP_TRY(hipMemcpy(host, dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return DSP_OK;
}
int dsp_h2d_async(void* dev, const void* host, int64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return DSP_OK;
}
int dsp_d2h_async(void* host, const void* dev, int64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return DSP_OK;
}
int dsp_stream_create(void** stream) {
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return DSP_OK;
}
int dsp_stream_destroy(void* stream) {
    HIP_RELEASE(hipStreamDestroy((hipStream_t)stream));
    return DSP_OK;
}
int dsp_stream_sync(void* stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return DSP_OK;
}
int dsp_sync(void) {
    HIP_TRY(hipDeviceSynchronize());
    return DSP_OK;
}
int dsp_event_create(void** event) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *event = (void*)e;
    return DSP_OK;
}
int dsp_event_destroy(void* event) {
    HIP_RELEASE(hipEventDestroy((hipEvent_t)event));
    return DSP_OK;
}
int dsp_event_record(void* event, void* stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return DSP_OK;
}
int dsp_stream_wait_event(void* stream, void* event) {
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return DSP_OK;
}
int dsp_host_register(void* host, int64_t bytes) {
    if (!host || bytes <= 0) return fail(DSP_ERR_ARG, "dsp_host_register: null pointer or empty range");
    hipError_t e = hipHostRegister(host, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // a refusal (range already registered, read-only mapping) must not poison the next launch check
        return fail(DSP_ERR_HIP, "hipHostRegister: %s", hipGetErrorString(e));
    }
    return DSP_OK;
}
int dsp_host_unregister(void* host) {
    if (!host) return DSP_OK;
    hipError_t e = hipHostUnregister(host);
    if (e != hipSuccess) {  // the caller must then keep the memory alive: the runtime still holds a record of the range
        (void)hipGetLastError();
        return fail(DSP_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(e));
    }
    return DSP_OK;
}
int dsp_event_sync(void* event) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return DSP_OK;
}
int dsp_event_elapsed_ms(void* start, void* stop, float* ms) {
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return DSP_OK;
}
// Diagnostics: the HIP / ROCr runtimes end the process with abort() on some failures (a GPU memory fault, an internal guarantee)
// without saying where.  With this handler installed SIGABRT first writes the native call stack to stderr, then takes its default
// course.  async-signal-safe calls only (backtrace_symbols_fd writes straight to the descriptor).
static struct sigaction g_prev_abort;
static bool g_abort_installed = false;
static int g_abort_fd = 2;
static void abort_trace_handler(int sig) {
    static const char head[] = "\n[dspeed_hip] SIGABRT -- native call stack of the aborting thread:\n";
    void* frames[64];
    const int n = backtrace(frames, 64);
    (void)!write(g_abort_fd, head, sizeof head - 1);
    backtrace_symbols_fd(frames, n, g_abort_fd);
    if (g_abort_fd != 2) {  // (a test runner may have redirected descriptor 2: say it there as well)
        (void)!write(2, head, sizeof head - 1);
        backtrace_symbols_fd(frames, n, 2);
    }
    // whoever was there before (Python's faulthandler prints its own stack), then the default; never this handler again -- a
    // re-raise into ourselves would loop for ever (the signal is blocked while we run and pends)
    struct sigaction next = g_prev_abort;
    if (!g_abort_installed || next.sa_handler == abort_trace_handler) {
        memset(&next, 0, sizeof next);
        next.sa_handler = SIG_DFL;
        sigemptyset(&next.sa_mask);
    }
    g_abort_installed = false;
    sigaction(sig, &next, nullptr);
    raise(sig);
}
int dsp_install_abort_trace(int fd) {
    g_abort_fd = fd >= 0 ? fd : 2;
    void* warm[2];
    (void)backtrace(warm, 2);  // (loads libgcc's unwinder now, not inside the handler)
    struct sigaction cur;
    if (sigaction(SIGABRT, nullptr, &cur) == 0 && cur.sa_handler == abort_trace_handler) return DSP_OK;  // already there: only the descriptor changed
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = abort_trace_handler;
    sigemptyset(&sa.sa_mask);
    struct sigaction prev;
    if (sigaction(SIGABRT, &sa, &prev) != 0) return fail(DSP_ERR_ARG, "sigaction(SIGABRT) failed");
    g_prev_abort = prev;  // the disposition of the FIRST installation is what an uninstall (or the handler) goes back to
    g_abort_installed = true;
    return DSP_OK;
}
int dsp_uninstall_abort_trace(void) {
    struct sigaction cur;
    if (!g_abort_installed || sigaction(SIGABRT, nullptr, &cur) != 0 || cur.sa_handler != abort_trace_handler) {
        g_abort_installed = false;  // (somebody else's handler is in place now: leave it)
        g_abort_fd = 2;
        return DSP_OK;
    }
    g_abort_installed = false;
    g_abort_fd = 2;
    return sigaction(SIGABRT, &g_prev_abort, nullptr) == 0 ? DSP_OK : fail(DSP_ERR_ARG, "sigaction(SIGABRT) failed");
}
const char* dsp_last_error(void) { return dsp_plan_last_error(); }
const char* dsp_version(void) { return "dspeed_hip 0.1 (gfx950)"; }


int dsp_chain_create(const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots,
                     int n_sregs, int compute_dtype, dsp_chain** out) {
    if (out) *out = nullptr;
    if (!out) return fail(DSP_ERR_ARG, "null out");
    std::unique_ptr<dsp_chain> ch(new dsp_chain());
    {  // everything that needs no device: validation, constants, LDS packing, kernel choice (dsp_plan.cpp)
        const int rc = dsp_plan_build(ch.get(), ops, n_ops, io, n_io, slot_len, n_slots, n_sregs, compute_dtype);
        if (rc != DSP_OK) return rc;
    }
    const DevProgram& P = ch->host;
    const int esz = (ch->f64 || ch->i64) ? 8 : 4;
    if (ch->fir_f16)  // device images of the float16 FIR's taps (rewritten by every launch: the taps are a binding)
        for (int k = 0; k < ch->fir.n_kernels; ++k) {
            void* buf = nullptr;
            HIP_TRY(hipMalloc(&buf, dsp_internal_fir_f16_taps_bytes(ch->fir.kend)));
            ch->f16.taps16[k] = buf;
        }
    HIP_TRY(hipGetDevice(&ch->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ch->device));
    ch->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // (the op array is the tail of the structure: only the ops the program has are allocated and uploaded -- 1.5 kB instead of 100 kB
    // for a one-processor chain)
    ch->dev_bytes = offsetof(DevProgram, ops) + (size_t)P.n_ops * sizeof(DevOp);
    HIP_TRY(hipMalloc((void**)&ch->dev, ch->dev_bytes));
    HIP_TRY(staged_h2d(ch->dev, &P, ch->dev_bytes));
    HIP_TRY(hipMalloc((void**)&ch->dev_err, DSP_ERR_WORDS * sizeof(int)));
    HIP_TRY(hipMemset(ch->dev_err, 0, DSP_ERR_WORDS * sizeof(int)));
    const int block_lds = ch->lds_bytes_per_wave * ch->waves_per_block;
    if (block_lds > 64 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_vm_lds(block_lds);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", block_lds, hipGetErrorString(e));
    }
    const int classic_lds = ch->lds_bytes_per_wave * ch->classic_wpb;
    if (ch->fused_ok && classic_lds > 64 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_energy_lds(ch->fused_trap, ch->fused_npf, classic_lds);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(energy kernel, %d): %s", classic_lds, hipGetErrorString(e));
    }
    if (ch->fir_f16 && dsp_internal_fir_f16_lds_bytes() > 64 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_fir_f16_lds(dsp_internal_fir_f16_lds_bytes());
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(float16 FIR kernel): %s", hipGetErrorString(e));
    }
    if (ch->fir_ok && ch->fir_lds_bytes > 64 * 1024) {
        hipError_t e = (hipError_t)(ch->fir.store ? dsp_internal_set_fir_store_lds(ch->fir_lds_bytes) : dsp_internal_set_fir_mfma_lds(ch->fir_lds_bytes));
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(FIR kernel, %d): %s", ch->fir_lds_bytes, hipGetErrorString(e));
    }
    if (ch->scalar_ok && n_sregs * 64 * esz > 48 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_scalar_lds(n_sregs * 64 * esz);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(scalar kernel): %s", hipGetErrorString(e));
    }
    if (ch->cur_ok && ch->cur_lds_bytes > 64 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_current_lds(ch->cur_lds_bytes);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(current kernel, %d): %s", ch->cur_lds_bytes, hipGetErrorString(e));
    }
    if (ch->rows_ok && ch->rows_lds_bytes > 64 * 1024) {
        hipError_t e = (hipError_t)dsp_internal_set_rows_lds(ch->rows_lds_bytes);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "hipFuncSetAttribute(rows kernel, %d): %s", ch->rows_lds_bytes, hipGetErrorString(e));
    }
    *out = ch.release();
    return DSP_OK;
}

// the lane-per-waveform kernel runs when the program has its shape and the row / coefficient buffers of this call keep 16-byte alignment
static bool rows_applies(const dsp_chain* ch, void* const* io_ptrs) {
    if (!ch->rows_ok || !ch->fused_on) return false;
    if (reinterpret_cast<uintptr_t>(io_ptrs[ch->rio_wf]) & 15u) return false;
    if (ch->rio_dwt >= 0 && ((reinterpret_cast<uintptr_t>(io_ptrs[ch->rio_dwt]) + 4u * (uintptr_t)ch->host.io[ch->rio_dwt].offset) & 15u)) return false;
    return true;
}

static bool fir_applies(const dsp_chain* ch, void* const* io_ptrs) {
    if (!ch->fir_ok || !ch->fused_on) return false;
    return (reinterpret_cast<uintptr_t>(io_ptrs[ch->fio_wf]) & 15u) == 0;
}

static int chain_blocks(const dsp_chain* ch, int64_t n_wf, int wpb, int cap_waves) {
    const int block_lds = ch->lds_bytes_per_wave * wpb;
    int per_cu = LDS_BYTES_PER_CU / block_lds;
    if (per_cu > cap_waves / wpb) per_cu = cap_waves / wpb;
    if (per_cu < 1) per_cu = 1;
    int64_t want = (n_wf + wpb - 1) / wpb;
    int64_t cap = (int64_t)ch->num_cu * per_cu;
    int64_t b = want < cap ? want : cap;
    return (int)(b > 0 ? b : 1);
}
static int vm_blocks(const dsp_chain* ch, int64_t n_wf) { return chain_blocks(ch, n_wf, ch->waves_per_block, ch->has_fir ? 8 : 12); }

// launch geometry of the register-resident kernel: up to 4 wavefronts per block, 2 wavefronts per SIMD (its register budget)
static void rr_geometry(const dsp_chain* ch, int64_t n_wf, int* wpb_out, int* blocks_out) {
    int wpb = LDS_BYTES_PER_CU / ch->rr_lds_bytes;
    if (wpb > 4) wpb = 4;
    if (wpb < 1) wpb = 1;
    if (const char* env = getenv("DSPEED_HIP_WPB")) {  // tuning knob: wavefronts per workgroup
        const int v = atoi(env);
        if (v >= 1 && v <= wpb) wpb = v;
    }
    int per_cu = LDS_BYTES_PER_CU / (ch->rr_lds_bytes * wpb);
    if (per_cu * wpb > 8) per_cu = 8 / wpb;
    if (per_cu < 1) per_cu = 1;
    const int64_t want = (n_wf + wpb - 1) / wpb, cap = (int64_t)ch->num_cu * per_cu;
    *wpb_out = wpb;
    *blocks_out = (int)(want < cap ? want : cap);
}

// launch geometry of the current-branch kernel (dsp_current.hip): persistent wavefronts, as many as a CU's LDS takes (at most 12 per CU) ...
static int current_blocks_cap(const dsp_chain* ch) {
    int per_cu = LDS_BYTES_PER_CU / ch->cur_lds_bytes;
    if (per_cu > 12) per_cu = 12;  // (three wavefronts per SIMD: the kernel's 109 - 163 registers allow them)
    return ch->num_cu * (per_cu < 1 ? 1 : per_cu);
}
// ... and of those as many as make every one walk the same number of 64-row groups (2 048 groups on 1 280 wavefronts are two rounds, the
// second three fifths empty; on 1 024 they are two full ones with a wavefront less per CU in each other's way).  One helper for the launch
// and for dsp_chain_geometry.
static int current_blocks(const dsp_chain* ch, int64_t n_wf) {
    const int64_t groups = (n_wf + 63) / 64, cap = current_blocks_cap(ch);
    const int64_t rounds = (groups + cap - 1) / cap;
    return (int)((groups + rounds - 1) / (rounds < 1 ? 1 : rounds));
}

// launch geometry of the run-length FIR (dsp_fir_runs.hip): persistent workgroups of four wavefronts, a row per wavefront and round; four
// workgroups per CU where LDS allows (the kernel's registers leave room for five wavefronts per SIMD), every one the same number of rounds
static int runs_blocks_cap(const dsp_chain* ch) {
    int per_cu = LDS_BYTES_PER_CU / dsp_internal_fir_runs_lds_bytes(ch->runs.m);
    if (per_cu > 4) per_cu = 4;
    return ch->num_cu * (per_cu < 1 ? 1 : per_cu);
}
static int runs_blocks(const dsp_chain* ch, int64_t n_wf) {
    const int64_t groups = (n_wf + 3) / 4, cap = runs_blocks_cap(ch);
    const int64_t rounds = (groups + cap - 1) / cap;
    return (int)((groups + rounds - 1) / (rounds < 1 ? 1 : rounds));
}

// makes `device` current for the calling thread and puts the previous one back when it goes out of scope
struct DeviceScope {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            ok = hipSetDevice(device) == hipSuccess;
            switched = ok;
        }
    }
    ~DeviceScope() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
};

static int post_err(dsp_chain* ch, void* stream) {
    if (ch->err_mirror) HIP_TRY(hipMemcpyAsync(ch->err_mirror, ch->dev_err, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return DSP_OK;
}

// the rows' scales and flags of a float16 FIR chain: grown to the largest batch seen
static int f16_rows_reserve(dsp_chain* ch, int64_t n_wf) {
    if (ch->f16_rows_cap >= n_wf) return DSP_OK;
    if (ch->f16.row_scale) HIP_TRY(hipFree(const_cast<void*>(ch->f16.row_scale)));
    if (ch->f16.row_flags) HIP_TRY(hipFree(const_cast<void*>(ch->f16.row_flags)));
    ch->f16.row_scale = ch->f16.row_flags = nullptr;
    ch->f16_rows_cap = 0;
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc(&a, (size_t)n_wf * sizeof(float)));
    ch->f16.row_scale = a;
    HIP_TRY(hipMalloc(&b, (size_t)n_wf * sizeof(unsigned)));
    ch->f16.row_flags = b;
    ch->f16_rows_cap = n_wf;
    return DSP_OK;
}

int dsp_chain_share_row_scales(dsp_chain* producer, dsp_chain* consumer) {
    if (!producer || !consumer || producer == consumer) return fail(DSP_ERR_ARG, "dsp_chain_share_row_scales: two chains");
    if (!producer->pz_ok || !consumer->fir_ok || !consumer->fir_f16 || producer->device != consumer->device || consumer->scale_feeder) return 0;
    if (consumer->fir.in_kind == 0 && consumer->fir.sub_mode == 0 && !producer->scale_sink) {  // the FIR reads the rows the kernel writes
        producer->scale_sink = consumer;
        consumer->scale_feeder = producer;
        consumer->fed_in_side = false;
        return 1;
    }
    // the FIR reads a slice of the rows the kernel reads, minus the same baseline column: integer rows only (a float32 row's NaN rule looks at the
    // samples around the slice as well); that the two are bound to the same buffers is checked at every execute
    const PzArgs& P = producer->pz;
    const FirArgs& F = consumer->fir;
    if (F.in_kind != 0 && F.in_kind == P.in_kind && F.sub_mode == 1 && P.sub_mode == 1 && consumer->fio_bl >= 0 && producer->pio_bl >= 0 &&
        F.wf_stride == P.wf_stride && F.bl_stride == P.bl_stride && F.wf_offset >= P.wf_offset && F.wf_offset + F.n <= P.wf_offset + P.len &&
        !producer->scale_sink_in) {
        producer->scale_sink_in = consumer;
        consumer->scale_feeder = producer;
        consumer->fed_in_side = true;
        return 1;
    }
    return 0;
}

int dsp_chain_execute(dsp_chain* ch, void* const* io_ptrs, int64_t n_wf, void* stream) {
    if (!ch || !io_ptrs) return fail(DSP_ERR_ARG, "null chain or io_ptrs");
    if (n_wf <= 0) return DSP_OK;
    IoPtrs ptrs{};
    for (int k = 0; k < ch->host.n_io; ++k) {
        if (!io_ptrs[k]) return fail(DSP_ERR_ARG, "io binding %d is NULL", k);
        ptrs.p[k] = io_ptrs[k];
    }
    DeviceScope on_chain_device(ch->device);  // the chain's program and error word live there; the caller's current device comes back on return
    if (!on_chain_device.ok) return fail(DSP_ERR_HIP, "hipSetDevice(%d) failed", ch->device);
    (void)hipGetLastError();  // launch checks below report this launch, not a stale error of an unrelated earlier call
    // a binding's first element: io_ptrs[k] + offset (the waveform input's offset travels in the kernels' arguments instead)
    auto at = [&](int k) -> void* {
        return k < 0 ? nullptr : (void*)((char*)io_ptrs[k] + (int64_t)ch->host.io[k].offset * elem_size(ch->host.io[k].dtype));
    };
    if (ch->scalar_ok && ch->fused_on) {
        hipError_t e = (hipError_t)dsp_internal_launch_scalar(ch->dev, &ptrs, n_wf, ch->host.n_sregs, ch->i64 ? 2 : (ch->f64 ? 1 : 0), (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "scalar kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (ch->pz_ok && ch->fused_on && (reinterpret_cast<uintptr_t>(io_ptrs[ch->pio_wf]) & 15u) == 0 && (reinterpret_cast<uintptr_t>(at(ch->pio_out)) & 15u) == 0) {
        PzArgs A = ch->pz;
        A.wf = io_ptrs[ch->pio_wf];
        A.bl = (const float*)at(ch->pio_bl);
        A.out = at(ch->pio_out);
        A.tau = (const float*)at(ch->pio_tau);
        for (int k = 0; k < 4; ++k) A.mm_out[k] = at(ch->pio_mm[k]);
        A.row_scale = nullptr;
        A.row_flags = nullptr;
        if (dsp_chain* sink = ch->scale_sink) {
            const int rc = f16_rows_reserve(sink, n_wf);
            if (rc != DSP_OK) return rc;
            A.row_scale = (float*)const_cast<void*>(sink->f16.row_scale);
            A.row_flags = (uint32_t*)const_cast<void*>(sink->f16.row_flags);
            sink->fed_rows_ptr = A.out;
            sink->fed_stream = stream;
            sink->fed_n_wf = n_wf;
            sink->fed_stride = A.out_stride;
            sink->fed_len = A.len;
        }
        A.in_scale = nullptr;
        A.in_flags = nullptr;
        if (dsp_chain* sink = ch->scale_sink_in) {
            const int rc = f16_rows_reserve(sink, n_wf);
            if (rc != DSP_OK) return rc;
            const int es = A.in_kind == 0 ? 4 : 2;
            A.in_scale = (float*)const_cast<void*>(sink->f16.row_scale);
            A.in_flags = (uint32_t*)const_cast<void*>(sink->f16.row_flags);
            A.in_lo = sink->fir.wf_offset - A.wf_offset;
            A.in_hi = A.in_lo + sink->fir.n;
            sink->fed_rows_ptr = (const char*)A.wf + (size_t)sink->fir.wf_offset * es;  // the consumer's first sample, if it is bound to these rows
            sink->fed_bl = A.bl;
            sink->fed_stream = stream;
            sink->fed_n_wf = n_wf;
            sink->fed_stride = A.wf_stride;
            sink->fed_len = sink->fir.n;
        }
        hipError_t e = (hipError_t)dsp_internal_launch_pz_rows(&A, n_wf, ch->dev_err, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "pole-zero rows kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (ch->red_ok && ch->fused_on) {
        ReduceArgs A = ch->red;
        A.wf = io_ptrs[ch->dio_wf];
        for (int k = 0; k < 5; ++k) A.out[k] = at(ch->dio_out[k]);
        for (int k = 0; k < DSP_REDUCE_PICKS; ++k) A.pick_out[k] = at(ch->dio_pick[k]);
        for (int k = 0; k < DSP_REDUCE_WALKS; ++k) {
            A.walk_out[k] = at(ch->dio_walk[k]);
            A.walk_thr[k] = (const float*)at(ch->dio_walk_thr[k]);
            A.walk_ts[k] = (const float*)at(ch->dio_walk_ts[k]);
        }
        const int vec = ch->red_vec && (reinterpret_cast<uintptr_t>(A.wf) & 15u) == 0;
        hipError_t e = (hipError_t)dsp_internal_launch_reduce(&A, n_wf, ch->red_dtype, vec, ch->dev_err, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "reduce kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (ch->runs_ok && ch->fused_on && (reinterpret_cast<uintptr_t>(io_ptrs[ch->uio_wf]) & 15u) == 0) {
        if (!ch->runs_table) HIP_TRY(hipMalloc((void**)&ch->runs_table, sizeof(FirRunsTable)));
        FirRunsArgs A = ch->runs;
        if (!A.keep && !ch->runs_scratch)
            HIP_TRY(hipMalloc((void**)&ch->runs_scratch, (size_t)runs_blocks_cap(ch) * 4 * (size_t)A.out_stride * sizeof(float)));
        A.wf = (const float*)io_ptrs[ch->uio_wf];
        A.taps = (const float*)at(ch->uio_taps);
        A.out = A.keep ? (float*)at(ch->uio_out) : ch->runs_scratch;
        A.table = ch->runs_table;
        for (int k = 0; k < 5; ++k) A.red.out[k] = at(ch->dio_out[k]);
        for (int k = 0; k < DSP_REDUCE_PICKS; ++k) A.red.pick_out[k] = at(ch->dio_pick[k]);
        for (int k = 0; k < DSP_REDUCE_WALKS; ++k) {
            A.red.walk_out[k] = at(ch->dio_walk[k]);
            A.red.walk_thr[k] = (const float*)at(ch->dio_walk_thr[k]);
            A.red.walk_ts[k] = (const float*)at(ch->dio_walk_ts[k]);
        }
        hipError_t e = (hipError_t)dsp_internal_launch_fir_runs(&A, ch->runs_table, n_wf, runs_blocks(ch, n_wf), ch->dev_err, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "run-length FIR kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (ch->cur_ok && ch->fused_on && (reinterpret_cast<uintptr_t>(io_ptrs[ch->cio_wf]) & 15u) == 0) {
        // persistent wavefronts, as many as a CU's LDS takes (at most 12 per CU): each keeps its scratch area for the groups of rows it walks
        const int cap = current_blocks_cap(ch);
        if (!ch->cur_scratch) {
            HIP_TRY(hipMalloc((void**)&ch->cur_scratch, (size_t)cap * (size_t)ch->cur.scratch_per_wave * sizeof(float)));
            ch->cur_blocks_cap = cap;
        }
        CurrentArgs A = ch->cur;
        A.wf = io_ptrs[ch->cio_wf];
        A.t0 = (const float*)at(ch->cio_t0);
        for (int k = 0; k < 4; ++k) A.out[k] = at(ch->cio_out[k]);
        A.scratch = ch->cur_scratch;
        const int blocks = current_blocks(ch, n_wf);
        hipError_t e = (hipError_t)dsp_internal_launch_current(&A, n_wf, blocks, ch->cur_lds_bytes, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "current kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (fir_applies(ch, io_ptrs)) {
        FirArgs A = ch->fir;
        A.wf = io_ptrs[ch->fio_wf];
        A.bl = (const float*)at(ch->fio_bl);
        for (int k = 0; k < A.n_kernels; ++k) {
            A.taps[k] = (const float*)at(ch->fio_taps[k]);
            A.out[k] = at(ch->fio_out[k]);
        }
        if (ch->fir_f16) {
            const int rc = f16_rows_reserve(ch, n_wf);
            if (rc != DSP_OK) return rc;
            // scales and flags already there?  Only if the kernel in front wrote exactly the rows this one reads, and just now
            const void* first = (const char*)A.wf + (size_t)A.wf_offset * (A.in_kind == 0 ? sizeof(float) : sizeof(int16_t));
            // -- and on this stream: scales and flags are written by the producer's launch, so only stream order puts them ahead of this one
            const bool same_batch = ch->scale_feeder && ch->fed_rows_ptr == first && ch->fed_n_wf == n_wf && ch->fed_stream == stream &&
                                    ch->fed_stride == A.wf_stride && ch->fed_len == A.n;
            ch->f16.rows_done = (same_batch && (ch->fed_in_side ? (A.in_kind != 0 && A.sub_mode == 1 && ch->fed_bl == (const void*)A.bl && A.bl != nullptr)
                                                                : (A.in_kind == 0 && A.sub_mode == 0))) ? 1 : 0;
            ch->fed_n_wf = -1;  // (a note is good for one execute)
        }
        hipError_t e = (hipError_t)(ch->fir_f16 ? dsp_internal_launch_fir_f16(&A, &ch->f16, n_wf, dsp_internal_fir_f16_lds_bytes(), (hipStream_t)stream)
                                    : A.store   ? dsp_internal_launch_fir_store(&A, n_wf, ch->fir_lds_bytes, (hipStream_t)stream)
                                                : dsp_internal_launch_fir_mfma(&A, n_wf, ch->fir_lds_bytes, (hipStream_t)stream));
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "FIR kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (rows_applies(ch, io_ptrs)) {
        RowsArgs A = ch->rows;
        A.wf = io_ptrs[ch->rio_wf];
        A.bl = (const float*)at(ch->rio_bl);
        A.thr = (const float*)at(ch->rio_thr);
        A.ts = (const float*)at(ch->rio_ts);
        for (int k = 0; k < 4; ++k) A.out_mm[k] = at(ch->rio_mm[k]);
        A.out_tpt = at(ch->rio_tpt);
        A.dwt_out = at(ch->rio_dwt);
        hipError_t e = (hipError_t)dsp_internal_launch_rows(&A, n_wf, ch->dev_err, ch->rows_lds_bytes, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "rows kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    const int blocks = vm_blocks(ch, n_wf);
    const int threads = 64 * ch->waves_per_block;
    const int lds = ch->lds_bytes_per_wave * ch->waves_per_block;
    if (ch->rr_ok && ch->fused_on && ch->variant != 1 && ((reinterpret_cast<uintptr_t>(io_ptrs[ch->io_wf]) & 15u) == 0)) {
        EnergyArgs F = ch->rr;
        F.wf = io_ptrs[ch->io_wf];
        F.bl = (const float*)at(ch->io_bl);
        F.tp = (const float*)at(ch->io_tp);
        F.tau = (const float*)at(ch->io_tau);
        F.out = (float*)at(ch->io_out);
        const int S = (ch->variant == 8 && ch->wf_dtype == DSP_F32) ? 2 : 1;
        int rwpb, rblocks;
        rr_geometry(ch, n_wf, &rwpb, &rblocks);
        hipError_t e = (hipError_t)dsp_internal_launch_energy_rr(&F, &ch->plan[S - 1], ch->fused_trap, ch->fused_npf, S, ch->wf_dtype, n_wf,
                                                                 ch->dev_err, rblocks, 64 * rwpb, ch->rr_lds_bytes * rwpb, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "energy kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    if (ch->fused_ok && ch->fused_on && ((reinterpret_cast<uintptr_t>(io_ptrs[ch->io_wf]) & 15u) == 0)) {
        EnergyArgs F = ch->fused;
        F.wf = io_ptrs[ch->io_wf];
        F.bl = (const float*)at(ch->io_bl);
        F.tp = (const float*)at(ch->io_tp);
        F.out = (float*)at(ch->io_out);
        const int cw = ch->classic_wpb;
        hipError_t e = (hipError_t)dsp_internal_launch_energy(&F, ch->fused_trap, ch->fused_npf, n_wf, ch->dev_err, chain_blocks(ch, n_wf, cw, 8),
                                                              64 * cw, ch->lds_bytes_per_wave * cw, (hipStream_t)stream);
        if (e != hipSuccess) return fail(DSP_ERR_HIP, "energy kernel launch failed: %s", hipGetErrorString(e));
        return post_err(ch, stream);
    }
    hipError_t e = ch->f64 ? (hipError_t)dsp_internal_launch_vm_f64(ch->dev, &ptrs, n_wf, ch->dev_err, blocks, threads, lds, ch->has_fir,
                                                                    (hipStream_t)stream)
                           : (hipError_t)dsp_internal_launch_vm_f32(ch->dev, &ptrs, n_wf, ch->dev_err, blocks, threads * ch->host.team, lds, ch->has_fir,
                                                                    ch->host.team, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DSP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return post_err(ch, stream);
}

int dsp_chain_check(dsp_chain* ch, void* stream, int64_t* row) {
    if (!ch) return fail(DSP_ERR_ARG, "null chain");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    int host_err[DSP_ERR_WORDS] = {0};
    if (ch->err_mirror) {
        memcpy(host_err, ch->err_mirror, 4 * sizeof(int));  // (copied by the launch itself, on its stream)
    } else {
        HIP_TRY(staged_d2h(host_err, ch->dev_err, sizeof host_err));
    }
#ifdef DSPEED_HIP_DIAG
    if (getenv("DSPEED_HIP_ABLATE") && (atoi(getenv("DSPEED_HIP_ABLATE")) & 8)) {  // diagnostic phase stamps
        unsigned long long ph[6];
        memcpy(ph, host_err + 4, sizeof ph);
        unsigned long long tot = 0;
        for (int i = 0; i < 6; ++i) tot += ph[i];
        if (tot) {
            fprintf(stderr, "[dspeed_hip stamps] cycles per phase (stage, pass1, pass2, carries, pass3, tail):");
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %.1f%%", 100.0 * (double)ph[i] / (double)tot);
            fprintf(stderr, "  total %llu\n", tot);
            HIP_TRY(hipMemset(ch->dev_err + 4, 0, sizeof ph));
        }
    }
#endif
    if (host_err[0] != 0) {
        if (row) *row = ((int64_t)(uint32_t)host_err[2] << 32) | (uint32_t)host_err[1];
        if (ch->err_mirror) memset(ch->err_mirror, 0, 4 * sizeof(int));  // consumed: a second check without a launch in between reports nothing
        HIP_TRY(hipMemsetAsync(ch->dev_err, 0, 4 * sizeof(int), (hipStream_t)stream));  // (in stream order, ahead of the next launch)
        dsp_set_last_error(dsp_fatal_message(host_err[0]));
        return host_err[0];
    }
    return DSP_OK;
}

int dsp_chain_profile(dsp_chain* ch, int enable) {
    if (!ch) return fail(DSP_ERR_ARG, "null chain");
    HIP_TRY(hipSetDevice(ch->device));
    const size_t bytes = (size_t)(ch->host.n_ops + 1) * sizeof(unsigned long long);
    if (enable) {
        if (!ch->host.prof) HIP_TRY(hipMalloc((void**)&ch->host.prof, bytes));
        HIP_TRY(hipMemset(ch->host.prof, 0, bytes));
    } else if (ch->host.prof) {
        HIP_RELEASE(hipFree(ch->host.prof));
        ch->host.prof = nullptr;
    }
    HIP_TRY(staged_h2d(ch->dev, &ch->host, ch->dev_bytes));
    return DSP_OK;
}

int dsp_chain_profile_read(dsp_chain* ch, int capacity, int32_t* opcodes, int32_t* slots, uint64_t* cycles, int* n_ops, uint64_t* n_waveforms) {
    if (!ch || !n_ops) return fail(DSP_ERR_ARG, "null argument");
    *n_ops = ch->host.n_ops;
    if (!ch->host.prof) return fail(DSP_ERR_ARG, "profiling is off: call dsp_chain_profile(chain, 1) and execute first");
    if (capacity < ch->host.n_ops || !opcodes || !cycles || !slots) return fail(DSP_ERR_ARG, "capacity %d < %d ops", capacity, ch->host.n_ops);
    HIP_TRY(hipSetDevice(ch->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> host(ch->host.n_ops + 1);
    HIP_TRY(staged_d2h(host.data(), ch->host.prof, host.size() * sizeof(unsigned long long)));
    for (int i = 0; i < ch->host.n_ops; ++i) {
        opcodes[i] = ch->host.ops[i].opcode;
        slots[i] = ch->host.ops[i].src;
        cycles[i] = host[i];
    }
    if (n_waveforms) *n_waveforms = host[ch->host.n_ops];
    return DSP_OK;
}

int dsp_chain_destroy(dsp_chain* ch) {
    if (!ch) return DSP_OK;
    delete ch;
    return DSP_OK;
}

int dsp_chain_geometry(dsp_chain* ch, int64_t n_wf, int* lds_bytes_per_wave, int* waves_per_block, int* blocks) {
    if (!ch) return fail(DSP_ERR_ARG, "null chain");
    if (ch->scalar_ok && ch->fused_on) {
        if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->host.n_sregs * 64 * ((ch->f64 || ch->i64) ? 8 : 4);
        if (waves_per_block) *waves_per_block = 1;
        if (blocks) *blocks = (int)((n_wf + 63) / 64);
        return DSP_OK;
    }
    if ((ch->red_ok || ch->pz_ok) && ch->fused_on) {
        if (lds_bytes_per_wave) *lds_bytes_per_wave = 0;
        if (waves_per_block) *waves_per_block = 4;
        if (blocks) *blocks = (int)((n_wf + 3) / 4);
        return DSP_OK;
    }
    if (ch->runs_ok && ch->fused_on) {
        if (lds_bytes_per_wave) *lds_bytes_per_wave = dsp_internal_fir_runs_lds_bytes(ch->runs.m) / 4;
        if (waves_per_block) *waves_per_block = 4;
        if (blocks) *blocks = runs_blocks(ch, n_wf);
        return DSP_OK;
    }
    if (ch->cur_ok && ch->fused_on) {
        if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->cur_lds_bytes;
        if (waves_per_block) *waves_per_block = 1;
        if (blocks) *blocks = current_blocks(ch, n_wf);
        return DSP_OK;
    }
    if (ch->fir_ok && ch->fused_on) {  // 8 wavefronts per 64 waveforms and kernel
        if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->fir_lds_bytes / 8;
        if (waves_per_block) *waves_per_block = 8;
        if (blocks) *blocks = (int)((n_wf + 63) / 64) * (ch->fir.store ? (ch->fir.p[0] + 319) / 320 : ch->fir.n_kernels);
        return DSP_OK;
    }
    if (ch->rows_ok && ch->fused_on) {  // a pair of wavefronts per 64 waveforms shares one history ring
        if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->rows_lds_bytes / 2;
        if (waves_per_block) *waves_per_block = 2;
        if (blocks) *blocks = (int)((n_wf + 63) / 64);
        return DSP_OK;
    }
    if (ch->rr_ok && ch->fused_on && ch->variant != 1) {
        int wpb, b;
        rr_geometry(ch, n_wf, &wpb, &b);
        if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->rr_lds_bytes;
        if (waves_per_block) *waves_per_block = wpb;
        if (blocks) *blocks = b;
        return DSP_OK;
    }
    if (lds_bytes_per_wave) *lds_bytes_per_wave = ch->lds_bytes_per_wave / ch->host.team;  // (a team of wavefronts shares a row's image)
    if (waves_per_block) *waves_per_block = ch->waves_per_block * ch->host.team;
    if (blocks) *blocks = vm_blocks(ch, n_wf);
    return DSP_OK;
}

const char* dsp_chain_kernel_name(dsp_chain* ch) { return dsp_plan_kernel_name(ch); }

const char* dsp_chain_kernel_note(dsp_chain* ch) {
    if (!ch) return "";
    const bool specialised = ch->fused_on && (ch->scalar_ok || ch->pz_ok || ch->red_ok || ch->runs_ok || ch->cur_ok || ch->fir_ok || ch->rows_ok || ch->rr_ok || ch->fused_ok);
    return specialised ? "" : ch->note.c_str();
}

int dsp_chain_set_async_check(dsp_chain* ch, int enable) {
    if (!ch) return fail(DSP_ERR_ARG, "null chain");
    if (enable && !ch->err_mirror) {
        HIP_TRY(hipHostMalloc((void**)&ch->err_mirror, DSP_ERR_WORDS * sizeof(int), hipHostMallocDefault));
        memset(ch->err_mirror, 0, DSP_ERR_WORDS * sizeof(int));
    } else if (!enable && ch->err_mirror) {
        (void)hipHostFree(ch->err_mirror);
        ch->err_mirror = nullptr;
    }
    return DSP_OK;
}

int dsp_chain_set_fused(dsp_chain* ch, int enable) {
    if (!ch) return fail(DSP_ERR_ARG, "null chain");
    ch->fused_on = (enable & 1) != 0;  // bit 0: use a specialised kernel
    // bits 1-3 pick one for cross-checks and A/B runs: 0 = default (register-resident where it applies), 6 = register-resident,
    // 7 = classic (VM layout); anything else = default
    const int v = (enable >> 1) & 7;
    ch->variant = (v == 7 || !ch->rr_ok) ? 1 : 6;
    return ((ch->fused_ok || ch->rr_ok || ch->rows_ok || ch->fir_ok || ch->cur_ok || ch->scalar_ok || ch->red_ok || ch->runs_ok || ch->pz_ok) && ch->fused_on) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ single processors
// One implementation per processor, generic in the loop type TY (DSP_F32 / DSP_F64); the exported dsp_<name>_f32 / _f64
// functions below only fix the scalar C types.
namespace {

struct MiniKey {
    std::vector<int64_t> v;
    bool operator<(const MiniKey& o) const { return v < o.v; }
};
std::map<MiniKey, dsp_chain*> g_cache;
std::mutex g_cache_mu;

int64_t dbits(double f) {
    int64_t i;
    memcpy(&i, &f, 8);
    return i;
}

struct Mini {
    int ty;  // DSP_F32 or DSP_F64: the gufunc loop
    std::vector<dsp_op> ops;
    std::vector<dsp_io_desc> io;
    std::vector<void*> ptrs;
    std::vector<int32_t> slots;
    int n_sregs = 0;
    MiniKey key;

    explicit Mini(int ty_) : ty(ty_) { key.v.push_back(ty_); }

    int add_io(int kind, int dtype, int len, int64_t stride, const void* p) {
        dsp_io_desc d{kind, dtype, len, 0, stride};
        io.push_back(d);
        ptrs.push_back(const_cast<void*>(p));
        key.v.insert(key.v.end(), {kind, dtype, len, stride});
        return (int)io.size() - 1;
    }
    int add_slot(int len) {
        key.v.push_back(len);
        slots.push_back(len);
        return (int)slots.size() - 1;
    }
    dsp_op& add_op(int opcode, int dst, int src, int io_idx) {
        dsp_op o;
        memset(&o, 0, sizeof o);
        o.opcode = opcode;
        o.dst = dst;
        o.src = src;
        o.io = io_idx;
        ops.push_back(o);
        key.v.insert(key.v.end(), {opcode, dst, src, io_idx});
        return ops.back();
    }
    // scalar gufunc argument: device column (of the loop type) if given, else broadcast constant
    dsp_scalar_arg scalar(const void* dev, double value) {
        dsp_scalar_arg a{DSP_ARG_CONST, 0, value};
        if (dev) {
            a.kind = DSP_ARG_INPUT;
            a.index = add_io(DSP_IO_SCALAR_IN, ty, 1, 1, dev);
        } else {
            key.v.push_back(dbits(value));
        }
        return a;
    }
    int run(int64_t n_wf, void* stream, int64_t* err_row) {
        for (auto& o : ops) {
            for (int k = 0; k < 4; ++k) key.v.push_back(o.ip[k]);
            key.v.push_back(o.io);
        }
        // The cached chains carry one device error word each and are destroyed when the cache is emptied: look-up, creation, execution and
        // the check of the error word happen under the cache's lock, so the dsp_<name>_f32/_f64 entry points are serialised within a
        // process (threads that want concurrency build their own chains with dsp_chain_create).
        std::lock_guard<std::mutex> lk(g_cache_mu);
        int dev = 0;
        (void)hipGetDevice(&dev);
        key.v.push_back(dev);
        dsp_chain* ch = nullptr;
        auto it = g_cache.find(key);
        if (it != g_cache.end()) ch = it->second;
        if (!ch) {
            int rc = dsp_chain_create(ops.data(), (int)ops.size(), io.data(), (int)io.size(), slots.data(), (int)slots.size(), n_sregs, ty,
                                      &ch);
            if (rc) return rc;
            if (g_cache.size() > 256) {  // (nobody is inside a cached chain: we hold the lock)
                for (auto& kv : g_cache) dsp_chain_destroy(kv.second);
                g_cache.clear();
            }
            g_cache[key] = ch;
        }
        int rc = dsp_chain_execute(ch, ptrs.data(), n_wf, stream);
        if (rc) return rc;
        return dsp_chain_check(ch, stream, err_row);
    }
};

struct WfIn {
    const void* ptr;
    int dtype;
    int64_t n_wf;
    int32_t len;
    int64_t stride;
};

// waveform -> waveform processors
int wf2wf(int ty, int opcode, const WfIn& in, void* out, int32_t out_len, int64_t out_stride, const int32_t* ip, int n_ip,
          const double* consts, int n_c, const void* col0, void* stream, int64_t* err_row, const void* const* cols = nullptr) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    // the per-sample and scan filters work in place: one LDS slot, so a wavefront holds a waveform twice as long (about 38 k float32
    // samples instead of 19 k for the filters that need source and destination side by side)
    const bool in_place = (opcode == DSP_OP_BL_SUBTRACT || opcode == DSP_OP_POLE_ZERO || opcode == DSP_OP_DOUBLE_POLE_ZERO) && out_len == in.len;
    const int s_in = m.add_slot(in.len), s_out = in_place ? s_in : m.add_slot(out_len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg sp[3];
    memset(sp, 0, sizeof sp);
    for (int k = 0; k < n_c; ++k) sp[k] = m.scalar(cols ? cols[k] : (k == 0 ? col0 : nullptr), consts[k]);  // device column or constant
    dsp_op& o = m.add_op(opcode, s_out, s_in, 0);
    for (int k = 0; k < n_ip; ++k) o.ip[k] = ip[k];
    for (int k = 0; k < n_c; ++k) o.sp[k] = sp[k];
    const int io_out = m.add_io(DSP_IO_WF_OUT, ty, out_len, out_stride, out);
    m.add_op(DSP_OP_STORE, 0, s_out, io_out);
    return m.run(in.n_wf, stream, err_row);
}

int g_bl_subtract(int ty, const WfIn& in, const void* bl_dev, double bl, void* out, int64_t out_stride, void* st, int64_t* er) {
    const double c[1] = {bl};
    return wf2wf(ty, DSP_OP_BL_SUBTRACT, in, out, in.len, out_stride, nullptr, 0, c, 1, bl_dev, st, er);
}
int g_min_max_norm(int ty, const WfIn& in, const void* lo_dev, double lo, const void* hi_dev, double hi, void* out, int64_t out_stride, void* st,
                   int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg a = m.scalar(lo_dev, lo), b = m.scalar(hi_dev, hi);
    dsp_op& o = m.add_op(DSP_OP_MIN_MAX_NORM, s_in, s_in, 0);  // in place
    o.sp[0] = a;
    o.sp[1] = b;
    const int io_out = m.add_io(DSP_IO_WF_OUT, ty, in.len, out_stride, out);
    m.add_op(DSP_OP_STORE, 0, s_in, io_out);
    return m.run(in.n_wf, st, er);
}
int g_pole_zero(int ty, const WfIn& in, const void* tau_dev, double tau, void* out, int64_t out_stride, void* st, int64_t* er) {
    const double c[1] = {tau};
    return wf2wf(ty, DSP_OP_POLE_ZERO, in, out, in.len, out_stride, nullptr, 0, c, 1, tau_dev, st, er);
}
int g_double_pole_zero(int ty, const WfIn& in, const void* const* cols, double tau1, double tau2, double frac, void* out, int64_t out_stride, void* st,
                       int64_t* er) {
    const double c[3] = {tau1, tau2, frac};
    return wf2wf(ty, DSP_OP_DOUBLE_POLE_ZERO, in, out, in.len, out_stride, nullptr, 0, c, 3, cols ? cols[0] : nullptr, st, er, cols);
}
int g_trap(int ty, int opcode, const WfIn& in, int32_t rise, int32_t flat, int32_t fall, void* out, int64_t out_stride, void* st, int64_t* er) {
    const int32_t ip[3] = {rise, flat, fall};
    return wf2wf(ty, opcode, in, out, in.len, out_stride, ip, 3, nullptr, 0, nullptr, st, er);
}
int g_dwt(int ty, const WfIn& in, int32_t level, int32_t coeff, void* out, int32_t out_len, int64_t out_stride, void* st, int64_t* er) {
    const int32_t ip[3] = {level, coeff, 0};  // scratch = the input slot itself (dead after the transform)
    return wf2wf(ty, DSP_OP_DWT_HAAR, in, out, out_len, out_stride, ip, 3, nullptr, 0, nullptr, st, er);
}
int g_convolve(int ty, const WfIn& in, const void* kernel_dev, int32_t kernel_len, int32_t mode, void* out, int32_t out_len, int64_t out_stride,
               void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    if (kernel_len <= 0) return fail(DSP_ERR_ARG, "empty kernel");
    // NaN among the taps -> NaN output (convolutions.py:45-46): look at them once on the host
    const size_t esz = ty == DSP_F64 ? 8 : 4;
    std::vector<unsigned char> taps(esz * (size_t)kernel_len);
    HIP_TRY(staged_d2h(taps.data(), kernel_dev, taps.size()));
    int has_nan = 0;  // bit 0: a NaN among the taps, bit 1: an infinity
    for (int k = 0; k < kernel_len; ++k) {
        const double v = ty == DSP_F64 ? ((const double*)taps.data())[k] : (double)((const float*)taps.data())[k];
        has_nan |= std::isnan(v) ? 1 : (std::isinf(v) ? 2 : 0);
    }
    Mini m(ty);
    const int s_in = m.add_slot(in.len), s_out = m.add_slot(out_len > 0 ? out_len : 1);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    const int io_k = m.add_io(DSP_IO_TAPS, ty, kernel_len, 0, kernel_dev);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_op& o = m.add_op(DSP_OP_CONVOLVE, s_out, s_in, io_k);
    o.ip[0] = mode;
    o.ip[1] = has_nan;
    const int io_out = m.add_io(DSP_IO_WF_OUT, ty, out_len > 0 ? out_len : 1, out_stride, out);
    m.add_op(DSP_OP_STORE, 0, s_out, io_out);
    return m.run(in.n_wf, st, er);
}
int g_pickoff(int ty, const WfIn& in, const void* t_dev, double t_in, int32_t mode, void* out, void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 1;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg t = m.scalar(t_dev, t_in);
    dsp_op& o = m.add_op(DSP_OP_PICKOFF, 0, s_in, 0);
    o.ip[0] = mode;
    o.sp[0] = t;
    const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, out);
    dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
    sto.ip[0] = 0;
    return m.run(in.n_wf, st, er);
}
// mode_char 0: time_point_thresh; otherwise interpolated_time_point_thresh with that interpolation mode
int g_tpt(int ty, const WfIn& in, const void* thr_dev, double thr, const void* ts_dev, double ts, double walk, void* out, void* st, int64_t* er,
          int mode_char = 0) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 1;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg a = m.scalar(thr_dev, thr), b = m.scalar(ts_dev, ts), c = m.scalar(nullptr, walk);
    dsp_op& o = m.add_op(mode_char ? DSP_OP_INTERP_TIME_POINT_THRESH : DSP_OP_TIME_POINT_THRESH, 0, s_in, 0);
    o.ip[0] = mode_char;
    o.sp[0] = a;
    o.sp[1] = b;
    o.sp[2] = c;
    const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, out);
    dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
    sto.ip[0] = 0;
    return m.run(in.n_wf, st, er);
}
int g_windower(int ty, const WfIn& in, const void* t0_dev, double t0, void* out, int32_t out_len, int64_t out_stride, void* st, int64_t* er) {
    const double c[1] = {t0};
    return wf2wf(ty, DSP_OP_WINDOWER, in, out, out_len, out_stride, nullptr, 0, c, 1, t0_dev, st, er);
}
int g_avg_current(int ty, const WfIn& in, double length, void* out, int32_t out_len, int64_t out_stride, void* st, int64_t* er) {
    const double c[1] = {length};
    return wf2wf(ty, DSP_OP_AVG_CURRENT, in, out, out_len, out_stride, nullptr, 0, c, 1, nullptr, st, er);
}
int g_trap_window_pickoff(int ty, const WfIn& in, int32_t rise, int32_t flat, const void* tp_dev, double tp, void* out, void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 1;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg a = m.scalar(tp_dev, tp);
    dsp_op& o = m.add_op(DSP_OP_TRAP_WINDOW_PICKOFF, 0, s_in, 0);
    o.ip[0] = rise;
    o.ip[1] = flat;
    o.sp[0] = a;
    const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, out);
    dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
    sto.ip[0] = 0;
    return m.run(in.n_wf, st, er);
}
int g_upsampler(int ty, const WfIn& in, double up, void* out, int32_t out_len, int64_t out_stride, void* st, int64_t* er) {
    const double c[1] = {up};
    return wf2wf(ty, DSP_OP_UPSAMPLER, in, out, out_len, out_stride, nullptr, 0, c, 1, nullptr, st, er);
}
int g_moving_window_multi(int ty, const WfIn& in, double length, double num_mw, int32_t mw_type, void* out, int64_t out_stride, void* st,
                          int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    if (floor(num_mw) != num_mw) return fail(DSP_E_MW_NUM_INT, "%s", dsp_fatal_message(DSP_E_MW_NUM_INT));
    Mini m(ty);
    const int s_in = m.add_slot(in.len), s_out = m.add_slot(in.len);
    const int s_tmp = num_mw > 1 ? m.add_slot(in.len) : 0;
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_op& o = m.add_op(DSP_OP_MOVING_WINDOW_MULTI, s_out, s_in, 0);
    o.ip[0] = mw_type;
    o.ip[1] = (int32_t)num_mw;
    o.ip[2] = s_tmp;
    o.sp[0] = m.scalar(nullptr, length);
    const int io_out = m.add_io(DSP_IO_WF_OUT, ty, in.len, out_stride, out);
    m.add_op(DSP_OP_STORE, 0, s_out, io_out);
    return m.run(in.n_wf, st, er);
}
int g_linear_slope_fit(int ty, const WfIn& in, void* mean, void* stdev, void* slope, void* intercept, void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 4;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    m.add_op(DSP_OP_LINEAR_SLOPE_FIT, 0, s_in, 0);
    void* outs[4] = {mean, stdev, slope, intercept};
    for (int k = 0; k < 4; ++k) {
        const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, outs[k]);
        dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
        sto.ip[0] = k;
    }
    return m.run(in.n_wf, st, er);
}
int g_mean_below(int ty, const WfIn& in, const void* thr_dev, double thr, void* out, void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 1;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    dsp_scalar_arg a = m.scalar(thr_dev, thr);
    dsp_op& o = m.add_op(DSP_OP_MEAN_BELOW, 0, s_in, 0);
    o.sp[0] = a;
    const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, out);
    dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
    sto.ip[0] = 0;
    return m.run(in.n_wf, st, er);
}
int g_min_max(int ty, const WfIn& in, void* t_min, void* t_max, void* a_min, void* a_max, void* st, int64_t* er) {
    if (in.n_wf <= 0) return DSP_OK;
    Mini m(ty);
    m.n_sregs = 4;
    const int s_in = m.add_slot(in.len);
    const int io_in = m.add_io(DSP_IO_WF_IN, in.dtype, in.len, in.stride, in.ptr);
    m.add_op(DSP_OP_LOAD, s_in, 0, io_in);
    m.add_op(DSP_OP_MIN_MAX, 0, s_in, 0);
    void* outs[4] = {t_min, t_max, a_min, a_max};
    for (int k = 0; k < 4; ++k) {
        const int io_out = m.add_io(DSP_IO_SCALAR_OUT, ty, 1, 1, outs[k]);
        dsp_op& sto = m.add_op(DSP_OP_STORE_SCALAR, 0, 0, io_out);
        sto.ip[0] = k;
    }
    return m.run(in.n_wf, st, er);
}

}  // namespace

#define DSP_GUFUNCS(SFX, TY, FT)                                                                                                              \
    int dsp_bl_subtract_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* baseline_dev,          \
                              FT baseline, FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                                     \
        return g_bl_subtract(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, baseline_dev, (double)baseline, out, out_stride, stream,        \
                             err_row);                                                                                                        \
    }                                                                                                                                         \
    int dsp_pole_zero_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT tau, FT* out,                   \
                            int64_t out_stride, void* stream, int64_t* err_row) {                                                             \
        return g_pole_zero(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, nullptr, (double)tau, out, out_stride, stream, err_row);          \
    }                                                                                                                                         \
    int dsp_pole_zero_col_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* tau_dev, FT tau,      \
                                FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                                                \
        return g_pole_zero(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, tau_dev, (double)tau, out, out_stride, stream, err_row);          \
    }                                                                                                                                         \
    int dsp_double_pole_zero_col_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* tau1_dev,     \
                                       FT tau1, const FT* tau2_dev, FT tau2, const FT* frac_dev, FT frac, FT* out, int64_t out_stride,        \
                                       void* stream, int64_t* err_row) {                                                                      \
        const void* cols[3] = {tau1_dev, tau2_dev, frac_dev};                                                                                 \
        return g_double_pole_zero(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, cols, (double)tau1, (double)tau2, (double)frac, out,      \
                                  out_stride, stream, err_row);                                                                              \
    }                                                                                                                                         \
    int dsp_double_pole_zero_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT tau1, FT tau2, FT frac,  \
                                   FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                                             \
        return g_double_pole_zero(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, nullptr, (double)tau1, (double)tau2, (double)frac, out,   \
                                  out_stride, stream, err_row);                                                                              \
    }                                                                                                                                         \
    int dsp_trap_filter_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,      \
                              FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                                                  \
        return g_trap(TY, DSP_OP_TRAP_FILTER, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, rise, flat, 0, out, out_stride, stream, err_row);  \
    }                                                                                                                                         \
    int dsp_trap_norm_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,        \
                            FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                                                    \
        return g_trap(TY, DSP_OP_TRAP_NORM, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, rise, flat, 0, out, out_stride, stream, err_row);    \
    }                                                                                                                                         \
    int dsp_asym_trap_filter_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat, \
                                   int32_t fall, FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                               \
        return g_trap(TY, DSP_OP_ASYM_TRAP, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, rise, flat, fall, out, out_stride, stream, err_row); \
    }                                                                                                                                         \
    int dsp_fixed_time_pickoff_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* t_in_dev,       \
                                     FT t_in, int32_t mode_char, FT* out, void* stream, int64_t* err_row) {                                   \
        return g_pickoff(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, t_in_dev, (double)t_in, mode_char, out, stream, err_row);           \
    }                                                                                                                                         \
    int dsp_time_point_thresh_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* threshold_dev,   \
                                    FT threshold, const FT* t_start_dev, FT t_start, FT walk_forward, FT* out, void* stream,                  \
                                    int64_t* err_row) {                                                                                       \
        return g_tpt(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, threshold_dev, (double)threshold, t_start_dev, (double)t_start,         \
                     (double)walk_forward, out, stream, err_row);                                                                             \
    }                                                                                                                                         \
    int dsp_interpolated_time_point_thresh_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,              \
                                                 const FT* threshold_dev, FT threshold, const FT* t_start_dev, FT t_start,                  \
                                                 int64_t walk_forward, int32_t mode_char, FT* out, void* stream, int64_t* err_row) {       \
        if (mode_char <= 0 || mode_char > 127) return fail(DSP_ERR_ARG, "interpolated_time_point_thresh: mode must be a character");       \
        return g_tpt(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, threshold_dev, (double)threshold, t_start_dev, (double)t_start,         \
                     (double)walk_forward, out, stream, err_row, mode_char);                                                                \
    }                                                                                                                                         \
    int dsp_min_max_norm_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* a_min_dev, FT a_min,  \
                               const FT* a_max_dev, FT a_max, FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                \
        return g_min_max_norm(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, a_min_dev, (double)a_min, a_max_dev, (double)a_max, out,       \
                              out_stride, stream, err_row);                                                                                  \
    }                                                                                                                                         \
    int dsp_windower_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* t0_dev, FT t0, FT* out,   \
                           int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row) {                                             \
        return g_windower(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, t0_dev, (double)t0, out, out_len, out_stride, stream, err_row);    \
    }                                                                                                                                         \
    int dsp_avg_current_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT length, FT* out,              \
                              int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row) {                                          \
        return g_avg_current(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, (double)length, out, out_len, out_stride, stream, err_row);     \
    }                                                                                                                                         \
    int dsp_trap_pickoff_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,     \
                               const FT* t_pickoff_dev, FT t_pickoff, FT* out, void* stream, int64_t* err_row) {                              \
        return g_trap_window_pickoff(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, rise, flat, t_pickoff_dev, (double)t_pickoff, out,      \
                                     stream, err_row);                                                                                       \
    }                                                                                                                                         \
    int dsp_upsampler_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT upsample, FT* out,              \
                            int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row) {                                            \
        return g_upsampler(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, (double)upsample, out, out_len, out_stride, stream, err_row);     \
    }                                                                                                                                         \
    int dsp_moving_window_multi_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT length, FT num_mw,    \
                                      int32_t mw_type, FT* out, int64_t out_stride, void* stream, int64_t* err_row) {                         \
        return g_moving_window_multi(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, (double)length, (double)num_mw, mw_type, out,          \
                                     out_stride, stream, err_row);                                                                           \
    }                                                                                                                                         \
    int dsp_linear_slope_fit_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT* mean, FT* stdev,       \
                                   FT* slope, FT* intercept, void* stream, int64_t* err_row) {                                                \
        return g_linear_slope_fit(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, mean, stdev, slope, intercept, stream, err_row);           \
    }                                                                                                                                         \
    int dsp_mean_below_threshold_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,                        \
                                       const FT* threshold_dev, FT threshold, FT* out, void* stream, int64_t* err_row) {                      \
        return g_mean_below(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, threshold_dev, (double)threshold, out, stream, err_row);         \
    }                                                                                                                                         \
    int dsp_min_max_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, FT* t_min, FT* t_max, FT* a_min,     \
                          FT* a_max, void* stream, int64_t* err_row) {                                                                        \
        return g_min_max(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, t_min, t_max, a_min, a_max, stream, err_row);                       \
    }                                                                                                                                         \
    int dsp_dwt_haar_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t level, int32_t coeff_char,  \
                           FT* out, int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row) {                                    \
        return g_dwt(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, level, coeff_char, out, out_len, out_stride, stream, err_row);          \
    }                                                                                                                                         \
    int dsp_convolve_wf_##SFX(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const FT* kernel_dev,            \
                              int32_t kernel_len, int32_t mode_char, FT* out, int32_t out_len, int64_t out_stride, void* stream,              \
                              int64_t* err_row) {                                                                                             \
        return g_convolve(TY, WfIn{in, in_dtype, n_wf, wf_len, in_stride}, kernel_dev, kernel_len, mode_char, out, out_len, out_stride,       \
                          stream, err_row);                                                                                                   \
    }

DSP_GUFUNCS(f32, DSP_F32, float)
DSP_GUFUNCS(f64, DSP_F64, double)
#undef DSP_GUFUNCS

int dsp_stream_read(const void* src, int64_t bytes, void* sink, void* stream) {
    if (!src || !sink || bytes < 16) return fail(DSP_ERR_ARG, "dsp_stream_read: need a source of at least 16 bytes and a 4-byte sink");
    if (reinterpret_cast<uintptr_t>(src) & 15u) return fail(DSP_ERR_ARG, "dsp_stream_read: source must be 16-byte aligned");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = (hipError_t)dsp_internal_launch_stream_read(src, bytes, (uint32_t*)sink, cus * 8, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DSP_ERR_HIP, "stream-read kernel launch failed: %s", hipGetErrorString(e));
    return DSP_OK;
}

int dsp_synth_waveforms(void* wf, int out_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, float* baseline, float* t_pick,
                        uint64_t seed, int64_t first_row, float tau, float sigma, float pick_offset, float bl_lo, float bl_hi,
                        float amp_lo, float amp_hi, void* stream) {
    if (out_dtype != DSP_F32 && out_dtype != DSP_I16) return fail(DSP_ERR_ARG, "synth output must be float32 or int16");
    if (n_wf <= 0) return DSP_OK;
    hipError_t e = (hipError_t)dsp_internal_launch_synth(wf, out_dtype, n_wf, wf_len, row_stride, baseline, t_pick, seed, first_row, tau,
                                                         sigma, pick_offset, bl_lo, bl_hi, amp_lo, amp_hi, 1.0f, 1.0f, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DSP_ERR_HIP, "synth launch failed: %s", hipGetErrorString(e));
    return DSP_OK;
}

int dsp_synth_pulses(void* wf, int out_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, float* baseline, float* t_pick, uint64_t seed,
                     int64_t first_row, float tau, float sigma, float pick_offset, float bl_lo, float bl_hi, float amp_lo, float amp_hi,
                     float rise_lo, float rise_hi, void* stream) {
    if (out_dtype != DSP_F32 && out_dtype != DSP_I16) return fail(DSP_ERR_ARG, "synth output must be float32 or int16");
    if (!(rise_lo >= 1.0f) || !(rise_hi >= rise_lo)) return fail(DSP_ERR_ARG, "synth: rise times are 1 <= rise_lo <= rise_hi samples");
    if (n_wf <= 0) return DSP_OK;
    hipError_t e = (hipError_t)dsp_internal_launch_synth(wf, out_dtype, n_wf, wf_len, row_stride, baseline, t_pick, seed, first_row, tau,
                                                         sigma, pick_offset, bl_lo, bl_hi, amp_lo, amp_hi, rise_lo, rise_hi, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DSP_ERR_HIP, "synth launch failed: %s", hipGetErrorString(e));
    return DSP_OK;
}

int dsp_linear_slope_fit_rows(const void* wf, int wf_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, int compute_dtype,
                              const void* sub_dev, int sub_dtype, double sub_const, int sub_mode, int has_pz, double pz_tau,
                              const dsp_fit_window* fits, int n_fits, void* out, void* stream) {
    if (compute_dtype != DSP_F32 && compute_dtype != DSP_F64) return fail(DSP_ERR_ARG, "compute_dtype must be DSP_F32 or DSP_F64");
    const bool f64 = compute_dtype == DSP_F64;
    if (!elem_size(wf_dtype) || wf_dtype == DSP_BOOL) return fail(DSP_ERR_ARG, "fit rows: unknown waveform dtype %d", wf_dtype);
    if (!f64 && (wf_dtype == DSP_I32 || wf_dtype == DSP_U32 || wf_dtype == DSP_F64))
        return fail(DSP_ERR_ARG, "fit rows: int32/uint32/float64 rows select the float64 loop (compute_dtype DSP_F64)");
    if (!fits || n_fits < 1 || n_fits > DSP_FIT_MAX) return fail(DSP_ERR_ARG, "fit rows: n_fits=%d out of range (1..%d)", n_fits, DSP_FIT_MAX);
    if (sub_mode < 0 || sub_mode > 2) return fail(DSP_ERR_ARG, "fit rows: sub_mode must be 0, 1 (bl_subtract) or 2 (numpy.subtract)");
    if (sub_mode && sub_dev && (!elem_size(sub_dtype) || sub_dtype == DSP_BOOL)) return fail(DSP_ERR_ARG, "fit rows: unknown dtype of the subtracted column");
    if (n_wf < 0 || wf_len <= 0 || row_stride < wf_len || (n_wf > 0 && (!wf || !out))) return fail(DSP_ERR_ARG, "fit rows: bad rows / buffers");
    FitArgs A;
    memset(&A, 0, sizeof A);
    A.wf = wf;
    A.n_wf = n_wf;
    A.row_stride = row_stride;
    A.wf_len = wf_len;
    A.sub = sub_mode ? sub_dev : nullptr;
    A.sub_dtype = sub_dtype;
    A.sub_mode = sub_mode;
    A.sub_const = f64 ? sub_const : (double)(float)sub_const;
    A.has_pz = has_pz ? 1 : 0;
    if (has_pz) {  // the constant as the chain's POLE_ZERO op forms it: the loop's scalar type, then exp(-1 / tau) in float64 through libm
        const double tau = f64 ? pz_tau : (double)(float)pz_tau;
        A.pz_nan = std::isnan(tau) ? 1 : 0;
        A.pz_c = std::exp(-1.0 / tau);
    }
    A.n_fits = n_fits;
    int max_end = 0;
    bool whole = sub_mode == 1;
    for (int k = 0; k < n_fits; ++k) {
        const dsp_fit_window& w = fits[k];
        if (w.stage < 0 || w.stage > 1 || (w.stage == 1 && !has_pz)) return fail(DSP_ERR_ARG, "fit rows: window %d: stage 1 is the pole-zero corrected waveform (has_pz)", k);
        if (w.first < 0 || w.count < 1 || (int64_t)w.first + w.count > wf_len) return fail(DSP_ERR_ARG, "fit rows: window %d is not inside the waveform", k);
        A.stage[k] = w.stage;
        A.first[k] = w.first;
        A.count[k] = w.count;
        if (w.first + w.count > max_end) max_end = w.first + w.count;
        whole |= w.stage == 1;
    }
    A.n_scan = whole ? wf_len : max_end;  // (the NaN rules of bl_subtract and pole_zero look at the whole waveform)
    A.out = out;
    if (n_wf == 0) return DSP_OK;
    hipError_t e = (hipError_t)dsp_internal_launch_fit_rows(&A, wf_dtype, compute_dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DSP_ERR_HIP, "fit rows launch failed: %s", hipGetErrorString(e));
    return DSP_OK;
}

}  // extern "C"
