/* dspeed_hip.h -- C ABI of the MI355X-native waveform DSP engine (libdspeed_hip.so).
 *
 * This is the drop-in boundary for dspeed's hot path:
 *   - the numba @guvectorize processors of the Ge energy chain
 *     (reference src/dspeed/processors/{bl_subtract,pole_zero,trap_filters,fixed_time_pickoff,
 *      time_point_thresh,min_max,dwt,convolutions}.py), and
 *   - the ProcessingChain inner loop that calls them block by block
 *     (reference src/dspeed/processing_chain.py:665-673 execute(), :1144-1163 _execute_procs(),
 *      :1778-1781 ProcessorManager.execute()).
 * The reference has no FFI of its own (it is pure Python over numba); the binding a maintainer adds is
 * the ctypes stub shown in INTEGRATION.md, which is what dspeed_amd/_lib.py implements.
 *
 * Conventions
 *   - plain pointers and sizes only; all `const void* dev` / `void* dev` pointers are DEVICE pointers
 *     unless the name says host;  waveforms are C-contiguous rows: row r starts at base + r*row_stride
 *     (in elements), exactly the (n_wf, wf_len) block layout of ProcChainVar buffers
 *     (processing_chain.py:259-269);
 *   - every function returns 0 (DSP_OK) or a negative DSP_ERR_* for API/runtime failures; positive
 *     DSP_E_* codes are the reference's DSPFatal conditions (config errors detected on the host at
 *     chain creation, data-dependent ones reported by dsp_chain_check);  nothing throws;
 *   - per-waveform failure is NaN output, never an error (docs/source/manuals/build_dsp.rst:152-175);
 *   - a chain handle is bound to the device current at creation and is not thread-safe; independent
 *     handles may be used from different threads/devices.  Global state: the thread-local string behind
 *     dsp_last_error(), and the cache of small chains behind the single-processor entry points
 *     (dsp_<name>_f32 / _f64), which is guarded by one lock -- those calls are serialised within a process.
 */
#ifndef DSPEED_HIP_H
#define DSPEED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------------------------- */
#define DSP_OK 0
#define DSP_ERR_HIP (-1)         /* a HIP runtime call failed (see dsp_last_error) */
#define DSP_ERR_ARG (-2)         /* malformed program / bad argument */
#define DSP_ERR_UNSUPPORTED (-3) /* valid in the reference but not implemented on the device path */
#define DSP_ERR_TOO_LONG (-4)    /* waveform does not fit the per-wavefront LDS budget */

/* DSPFatal conditions of the reference processors (message texts: dsp_fatal_message) */
#define DSP_E_PZ_NAN 1         /* pole_zero.py:76-77   data dependent */
#define DSP_E_DPZ_SHORT 2      /* pole_zero.py:163-166 */
#define DSP_E_TRAP_RISE 3      /* trap_filters.py:53-54 */
#define DSP_E_TRAP_FLAT 4      /* trap_filters.py:56-57 */
#define DSP_E_TRAP_FALL 5      /* trap_filters.py:205-206 */
#define DSP_E_TRAP_WIDE 6      /* trap_filters.py:59-60 */
#define DSP_E_FTP_INT 7        /* fixed_time_pickoff.py:84-85   data dependent */
#define DSP_E_FTP_MODE 8       /* fixed_time_pickoff.py:124-125 data dependent (only raised for non-integer t_in) */
#define DSP_E_TPT_START_INT 9  /* time_point_thresh.py:67-68    data dependent */
#define DSP_E_TPT_WALK_INT 10  /* time_point_thresh.py:70-71 */
#define DSP_E_TPT_RANGE 11     /* time_point_thresh.py:73-74    data dependent */
#define DSP_E_CONV_LONG 12     /* convolutions.py:48-49 */
#define DSP_E_CONV_OUTLEN 13   /* convolutions.py:52-67 */
#define DSP_E_CONV_MODE 14     /* convolutions.py:69-70 */
#define DSP_E_DWT_LEVEL 15     /* dwt.py:67-68 */
#define DSP_E_DWT_OUTLEN 16    /* shape mismatch in dwt.py:81 */
#define DSP_E_ZERODIV 17       /* numba error_model='python': division by a zero rise/fall */
#define DSP_E_WINDOW_LONG 18   /* windower.py:36-37 */
#define DSP_E_AVGCUR_RANGE 19  /* moving_windows.py:243-246 */
#define DSP_E_TPO_INT 20       /* trap_filters.py:270-271       data dependent */
#define DSP_E_UPSAMPLE 21      /* upsampler.py:47-48 */
#define DSP_E_MW_LEN_INT 22    /* moving_windows.py:167-168 */
#define DSP_E_MW_NUM_INT 23    /* moving_windows.py:170-171 */
#define DSP_E_MW_LEN_RANGE 24  /* moving_windows.py:173-174 */
#define DSP_E_MW_NUM_NEG 25    /* moving_windows.py:176-177 */

/* ---- element types -------------------------------------------------------------------------- */
#define DSP_F32 0
#define DSP_F64 1
#define DSP_I16 2
#define DSP_U16 3
#define DSP_I32 4
#define DSP_U32 5
#define DSP_BOOL 6 /* one byte per element, 0 / 1: outputs (results of comparisons, isnan, isfinite) and per-event input columns */
#define DSP_I64 7  /* per-event columns, and the compute type of integer programs (below) */
#define DSP_U64 8  /* per-event columns */

/* ---- device, memory, streams (thin, so a host needs nothing but this library) --------------------- */
int dsp_device_count(int* count);
int dsp_set_device(int device);
int dsp_get_device(int* device);
int dsp_device_info(int device, char* name, int name_cap, int* compute_units, int64_t* hbm_bytes, int* lds_bytes_per_cu);
int dsp_malloc(void** dev, int64_t bytes);
int dsp_free(void* dev);
int dsp_host_alloc(void** host, int64_t bytes); /* pinned, for async staging */
int dsp_host_free(void* host);
int dsp_host_register(void* host, int64_t bytes);   /* page-lock caller-owned memory (e.g. a NumPy buffer) in place: async copies then */
int dsp_host_unregister(void* host);                /* overlap kernels without a staging memcpy; undo before the memory is freed */
int dsp_memset(void* dev, int value, int64_t bytes, void* stream);
int dsp_h2d(void* dev, const void* host, int64_t bytes);                      /* synchronous */
int dsp_d2h(void* host, const void* dev, int64_t bytes);                      /* synchronous */
int dsp_h2d_async(void* dev, const void* host, int64_t bytes, void* stream);  /* host must be pinned to overlap */
int dsp_d2h_async(void* host, const void* dev, int64_t bytes, void* stream);
int dsp_stream_create(void** stream);
int dsp_stream_destroy(void* stream);
int dsp_stream_sync(void* stream); /* NULL = default stream */
int dsp_sync(void);                /* whole device */
int dsp_event_create(void** event);
int dsp_event_destroy(void* event);
int dsp_event_record(void* event, void* stream);
int dsp_stream_wait_event(void* stream, void* event); /* work queued on `stream` after this call waits for `event` */
int dsp_event_sync(void* event);
int dsp_event_elapsed_ms(void* start, void* stop, float* ms);
int dsp_install_abort_trace(int fd);           /* diagnostics: on SIGABRT (how the HIP / ROCr runtimes end the process on a GPU fault)
                                               * write the native call stack to descriptor fd (< 0: stderr) before the default action;
                                               * calling it again only changes the descriptor */
int dsp_uninstall_abort_trace(void);           /* put back the SIGABRT disposition found by the first dsp_install_abort_trace */
const char* dsp_last_error(void);             /* thread-local text of the last failure */
const char* dsp_fatal_message(int dsp_e_code); /* the reference's DSPFatal message for a DSP_E_* code */
const char* dsp_version(void);

/* ---- fused chains: the ProcessingChain inner loop on the device ------------------------------------
 * A chain is a small program run by one wavefront per waveform with every intermediate waveform
 * resident in LDS.  It replaces, for one batch of n_wf rows, the reference's
 *     for block in range(0, n, 16): read inputs; for proc in procs: proc.execute(); write outputs
 * (processing_chain.py:665-673, 1144-1163).  The host (dspeed_amd.processing_chain) translates a
 * dspeed JSON recipe into this program.
 */
#define DSP_MAX_OPS 192   /* a whole LEGEND recipe (tests/configs/icpc-dsp-config.json: 43 processors, 34 outputs) is one program */
#define DSP_MAX_SLOTS 32  /* waveform variables; slots whose lifetimes do not overlap share LDS (packed by dsp_chain_create) */
#define DSP_MAX_IO 128 /* (a whole recipe's scalar tail binds its 34 outputs, the registers handed over to it and the per-row offsets of its grids) */
#define DSP_MAX_SREGS 128

/* I/O binding kinds */
#define DSP_IO_WF_IN 0      /* waveform input  (n_wf rows of `len` samples starting at `offset` within each row) */
#define DSP_IO_WF_OUT 1     /* waveform output */
#define DSP_IO_SCALAR_IN 2  /* one value per waveform */
#define DSP_IO_SCALAR_OUT 3 /* one value per waveform */
#define DSP_IO_TAPS 4       /* one constant vector shared by all waveforms (FIR kernel), len elements */

typedef struct dsp_io_desc {
    int32_t kind;       /* DSP_IO_* */
    int32_t dtype;      /* DSP_F32 ... ; outputs have the chain's compute type, or DSP_BOOL (nonzero -> 1); any type in an integer program */
    int32_t len;        /* samples per row used by the chain (1 for scalars) */
    int32_t offset;     /* first sample within the row: a constant slice wf[offset:offset+len] costs nothing */
    int64_t row_stride; /* elements between consecutive rows (>= offset+len; 1 for scalars; 0 = same value for all rows) */
} dsp_io_desc;

/* scalar operand of an op */
#define DSP_ARG_CONST 0 /* value */
#define DSP_ARG_INPUT 1 /* index = I/O binding of kind DSP_IO_SCALAR_IN */
#define DSP_ARG_REG 2   /* index = scalar register written by an earlier op */
typedef struct dsp_scalar_arg {
    int32_t kind;
    int32_t index;
    double value;
} dsp_scalar_arg;

/* opcodes; (reference processor, file:line) */
#define DSP_OP_LOAD 1          /* dst <- io (waveform input; int16/uint16 rows are widened like NumPy's ufunc casting, processing_chain.py:1565-1572);
                                  * ip[0] / ip[1] > 0: the binding is a slice of a longer waveform whose first consumer is a processor with the
                                  * "NaN anywhere -> NaN waveform" rule applied to the WHOLE waveform (bl_subtract.py:41-44): the ip[0] samples
                                  * before and the ip[1] samples after the slice are screened too, a NaN there marks the slot NaN;
                                  * ip[2] bit 0: a promise about the rows -- a NaN anywhere in a row means every sample of it is NaN (what
                                  * pole_zero writes): a kernel that reads only part of a row need not screen the rest of it */
#define DSP_OP_STORE 2         /* io <- src */
#define DSP_OP_STORE_SCALAR 3  /* io <- sreg[ip[0]] */
#define DSP_OP_BL_SUBTRACT 4   /* bl_subtract.py:11-46      dst <- src - sp[0]; ip[0] = 1: numpy.subtract(w, scalar), NaN samples stay single */
#define DSP_OP_POLE_ZERO 5     /* pole_zero.py:24-77        sp[0] = tau (const) */
#define DSP_OP_DOUBLE_POLE_ZERO 6 /* pole_zero.py:82-198    sp[0..2] = tau1, tau2, frac (const) */
#define DSP_OP_TRAP_FILTER 7   /* trap_filters.py:12-76     ip[0..1] = rise, flat */
#define DSP_OP_TRAP_NORM 8     /* trap_filters.py:79-149 */
#define DSP_OP_ASYM_TRAP 9     /* trap_filters.py:152-227   ip[0..2] = rise, flat, fall */
#define DSP_OP_PICKOFF 10      /* fixed_time_pickoff.py:12-125  sreg[dst] <- src at sp[0]; ip[0] = mode char; ip[1] = 1: the sample src[sp[0]] itself (wf[i] in a recipe);
                                  * ip[1] = 2: get.py:50-92 get_default -- src[int(sp[0])] with a per-event index (negative: from the end), sp[1] where the
                                  * index is outside the array or the sample is NaN (wf[variable] in a recipe, processing_chain.py:991-1005) */
#define DSP_OP_TIME_POINT_THRESH 11 /* time_point_thresh.py:12-92  sreg[dst] <- src; sp[0..2] = threshold, t_start, walk_forward */
#define DSP_OP_MIN_MAX 12      /* min_max.py:11-82          sreg[dst..dst+3] <- t_min, t_max, a_min, a_max */
#define DSP_OP_DWT_HAAR 13     /* dwt.py:13-81              dst <- src; ip[0] = level, ip[1] = 'a'|'d', ip[2] = scratch slot */
#define DSP_OP_CONVOLVE 14     /* convolutions.py:14-72,75-119  dst <- src (*) io taps; ip[0] = mode char f|v|s, ip[1] = what the caller
                                  * found among the taps: bit 0 a NaN (output NaN), bit 1 an infinity; ip[3] > 0: the kernel has ip[3]
                                  * taps and the binding (longer, ideally a multiple of 16) holds zeros after them -- lets the blocked tap
                                  * loop cover every tap; ip[2] = 1: the caller found the kernel piecewise constant (kernels.py t0_filter,
                                  * moving averages: a few runs of equal taps) -- a hint only, the kernel in use looks at the taps of
                                  * every launch itself: LOAD, CONVOLVE, [STORE], [per-event reductions of the filtered waveform] then
                                  * runs on prefix sums (dsp_fir_runs_kernel) */
#define DSP_OP_COPY 15         /* dst[k] <- src[ip[0] + k * step], step = ip[1] (0 stands for 1; negative: backwards)  (constant slice of an intermediate: processing_chain.py:1009-1071) */
#define DSP_OP_TRAP_PICKOFF 16 /* fusion of TRAP_FILTER|TRAP_NORM|ASYM_TRAP (ip[3] = which opcode) with PICKOFF: the trap output is
                                  never materialised; sreg[dst] <- trap(src) at sp[0]; ip[0..2] = rise, flat, fall; mode in `io` */
#define DSP_OP_AMAX 17         /* numpy.amax along the sample axis (icpc-dsp-config.json:123-143): sreg[dst] <- max(src), NaN if any NaN */
#define DSP_OP_MEAN_BELOW 19    /* arithmetic.py:9-62 mean_below_threshold: sreg[dst] <- mean of the samples of src below sp[0]; NaN if none */
#define DSP_OP_CONVOLVE_AMAX 20  /* fusion of CONVOLVE with numpy.amax over its output (icpc-dsp-config.json:160-239 cuspEmax / zacEmax): the
                                  * filtered waveform is never stored.  sreg[dst] <- max_o (src (*) io taps)[o]; ip[0] = mode, ip[1] = taps
                                  * hold a NaN, ip[2] = output length the recipe declared */
#define DSP_OP_WINDOWER 21      /* windower.py:12-54         dst[k] <- src[int(sp[0]) + k], NaN where that falls outside src */
#define DSP_OP_AVG_CURRENT 22   /* moving_windows.py:206-249 dst[k] <- (src[k + L] - src[k]) / sp[0], L = int(sp[0]) (constant) */
#define DSP_OP_TRAP_WINDOW_PICKOFF 23 /* trap_filters.py:230-293 trap_pickoff: sreg[dst] <- (sum of the rise samples ending at sp[0]
                                  * minus the rise samples ending rise+flat earlier) / rise; ip[0..1] = rise, flat */
#define DSP_OP_TRAP_REDUCE 24    /* fusion of TRAP_FILTER|TRAP_NORM|ASYM_TRAP (ip[3]) with the min_max and / or time_point_thresh that are its only
                                  * consumers: the filtered waveform is never stored.  dst = first of 4 registers t_min,t_max,a_min,a_max
                                  * (or -1), io = time_point_thresh register (or -1), sp[0..2] = threshold, t_start, walk_forward.
                                  * ip[3] bits 0-7 = the trapezoid's opcode; bits 8-15 = mode char of a fixed_time_pickoff that also reads
                                  * the trapezoid (0: none; not 's'), at sp[3], into register (bits 16-29) - 1; bit 30: only a_max of the
                                  * four values is wanted (numpy.amax: trapEmax + trapEftp of the Ge recipes in one pass) */
#define DSP_OP_UPSAMPLER 25      /* upsampler.py:13-56        dst <- every sample of src repeated int(sp[0]) times (constant factor), NaN where nothing lands */
#define DSP_OP_MOVING_WINDOW_MULTI 26 /* moving_windows.py:117-204  dst <- ip[1] moving averages of src, length sp[0] (constant), ip[0] = mw_type,
                                  * ip[2] = scratch slot (needed for two or more windows; with an odd number of windows it may be src
                                  * itself, which is then overwritten).  ip[3] = 1: in place -- dst is src, every pass overwrites it, ip[2]
                                  * is a side slot of 64 x window samples (the ends of the lanes' chunks); the window must fit one chunk */
#define DSP_OP_LINEAR_SLOPE_FIT 27 /* linear_slope_fit.py:11-91  sreg[dst..dst+3] <- mean, stdev (Welford, in the reference's rounding
                                  * sequence), slope, intercept of src[ip[0] : ip[0] + ip[1]] (ip[1] == 0: to the end of the slot) */
#define DSP_OP_SCALAR_AFFINE 18 /* sreg[dst] <- sp[0] * sp[1] + sp[2]  (recipe expressions: tp_0 + 10*us, 0.9*trapTmax, a + b, a * b, a - b) */
#define DSP_OP_SCALAR_CONVERT 28 /* unit_conversion.py:16-79  sreg[dst] <- f((sp[0] + sp[1]) * sp[3] - sp[2]) in float64, rounded to the loop type:
                                  * a time coordinate moved between two CoordinateGrids (processing_chain.py:1806-1908).  sp[1] / sp[2] = offset of
                                  * the source / target grid in periods, sp[3] = ratio of the periods (constant), ip[0] = f: 0 none, 1 rint,
                                  * 2 floor, 3 ceil, 4 trunc (round()/floor()/ceil()/trunc() onto a grid, processing_chain.py:1193-1266) */
#define DSP_OP_INTERP_TIME_POINT_THRESH 30 /* time_point_thresh.py:95-222 interpolated_time_point_thresh: sreg[dst] <- crossing of sp[0] found from
                                  * sp[1] walking forward (sp[2] > 0) or backward, placed between the two samples by ip[0] = mode char
                                  * i b c a f r n l; a start outside the waveform gives NaN (no DSPFatal) */
#define DSP_OP_MIN_MAX_NORM 31   /* min_max.py:85-140 min_max_norm: dst <- src / max(|sp[0]|, |sp[1]|) (a_min, a_max); src unchanged if either
                                  * is 0; NaN if src has a NaN or a bound is NaN */
#define DSP_OP_SCALAR_DIV 29     /* sreg[dst] <- sp[0] / sp[1]  (numpy.true_divide between per-event variables: QDrift / trapTmax) */
#define DSP_OP_ELEMENTWISE 32    /* NumPy ufuncs the recipe language adds as processors (processing_chain.py:832-947 operators and comparisons,
                                 * :1266-1420 astype / isnan / isfinite / where), sample by sample: dst[k] <- f(A[k], B[k], C[k]); ip[0] = DSP_FN_*;
                                 * operand A = waveform slot src if src >= 0 else sp[0], B = slot ip[1] if >= 0 else sp[1], C = slot ip[2] if >= 0
                                 * else sp[2] (at least one operand is a slot; all of the length of dst); truth values are 0 / 1 in the loop type */
#define DSP_OP_SCALAR_FUNC 33    /* the same functions between per-event values: sreg[dst] <- f(sp[0], sp[1], sp[2]); ip[0] = DSP_FN_* */
#define DSP_FN_ADD 0
#define DSP_FN_SUB 1
#define DSP_FN_MUL 2
#define DSP_FN_DIV 3
#define DSP_FN_LT 4
#define DSP_FN_LE 5
#define DSP_FN_GT 6
#define DSP_FN_GE 7
#define DSP_FN_EQ 8
#define DSP_FN_NE 9
#define DSP_FN_WHERE 10    /* A != 0 ? B : C */
#define DSP_FN_ISNAN 11
#define DSP_FN_ISFINITE 12
#define DSP_FN_NEG 13
#define DSP_FN_COPY 14     /* astype to the loop type */
#define DSP_FN_FLOORDIV 15 /* numpy.floor_divide's float loops (npy_divmod: the quotient from fmod, so that it never lies above A / B's true floor) */
/* NumPy's INTEGER ufunc loops (the first signature every operand can be cast to is an integer one when all operands are integer columns,
 * processing_chain.py:1565-1572, 1654-1664): the operands hold integers exactly in the loop type, the operation is done on 64-bit integers
 * and the result wrapped to the loop's integer type, the way 'hh->h', 'HH->H', 'ii->i' ... do.  ip[0] = DSP_FN_I* | DSP_FN_INT(bits, signed)
 * with bits 8, 16 or 32 (32 needs the float64 chain: a float32 does not hold every int32), or 64 in the float64 chain for IADD / ISUB / IMUL /
 * IFLOORDIV when the caller knows that operands and result stay below 2^53 in magnitude (int32 beside uint32 samples: nothing wraps there and
 * every value is a float64; 64-bit loops between per-event values run exactly in an integer program, below). */
#define DSP_FN_IADD 16
#define DSP_FN_ISUB 17
#define DSP_FN_IMUL 18
#define DSP_FN_IFLOORDIV 19 /* floor division, 0 where B == 0 (numpy.floor_divide's integer loops) */
#define DSP_FN_ICAST 20     /* astype to an integer type (:1268-1300, numpy.copyto(casting="unsafe")): truncation towards zero, then the wrap; a value the
                             * C conversion does not define (NaN, beyond the 32-bit / 64-bit range it goes through) gives what x86-64's cvtt* gives */
#define DSP_FN_LOR 21       /* A != 0 || B != 0: numpy.add's '??->?' loop, what the language's + is between truth values (processing_chain.py:832-891) */
#define DSP_FN_LAND 22      /* A != 0 && B != 0: numpy.multiply's '??->?' loop */
#define DSP_FN_RINT 23      /* round / floor / ceil / trunc of every sample (processors/round_to_nearest.py: the language's round(wf, to_nearest) is */
#define DSP_FN_FLOOR 24     /* A / to_nearest, one of these, times to_nearest) */
#define DSP_FN_CEIL 25
#define DSP_FN_TRUNC 26
#define DSP_FN_LAST 26
/* INTEGER PROGRAMS (compute_dtype DSP_I64): arithmetic between per-event INTEGER values whose NumPy loop is a 64-bit one ('ll->l', 'QQ->Q':
 * int64 / uint64 columns, int32 beside uint32) cannot be held in a float loop type.  A program of SCALAR_FUNC and STORE_SCALAR ops only may
 * be created with compute_dtype DSP_I64: its registers are 64-bit integers, input columns are integer or DSP_BOOL columns read exactly,
 * DSP_FN_IADD ... DSP_FN_ICAST take bits 8 / 16 / 32 / 64 and wrap to that type as NumPy's loops do (two's complement; floor division by 0
 * gives 0), comparisons / WHERE / LOR / LAND / COPY work on the integers -- a comparison or floor division of uint64 values carries
 * DSP_FN_INT(64, 0) and is done unsigned --, and a STORE_SCALAR writes the binding's own type: DSP_I64 / DSP_U64 as they are, narrower integers
 * truncated, DSP_BOOL as != 0, DSP_F32 / DSP_F64 converted (ip[1] = 1: the register holds a uint64).  dspeed_amd's recipe builder runs such a
 * program ahead of the main one and hands its columns on. */
#define DSP_FN_INT(bits, is_signed) (((bits) << 8) | ((is_signed) ? 1 << 16 : 0))
#define DSP_FN_CODE(ip0) ((ip0) & 0xff)
#define DSP_FN_INT_BITS(ip0) (((ip0) >> 8) & 0xff)
#define DSP_FN_INT_SIGNED(ip0) (((ip0) >> 16) & 1)

typedef struct dsp_op {
    int32_t opcode;
    int32_t dst; /* waveform slot, or first scalar register written */
    int32_t src; /* waveform slot read */
    int32_t io;  /* I/O binding index (LOAD/STORE/STORE_SCALAR/CONVOLVE) or mode char (TRAP_PICKOFF) */
    int32_t ip[4];
    dsp_scalar_arg sp[4];
} dsp_op;

typedef struct dsp_chain dsp_chain; /* opaque */

/* compute_dtype: DSP_F32 (the loop int16/uint16/float32 inputs select), DSP_F64 (float64/int32/uint32 inputs) or DSP_I64 (an integer program of
 * per-event values, above).
 * slot_len[s] = number of samples held by waveform slot s (static per chain, like ProcChainVar shapes).
 * Validates the program and every constant-only DSPFatal condition; on failure returns the code and *out = NULL. */
int dsp_chain_create(const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots,
                     int n_sregs, int compute_dtype, dsp_chain** out);
/* What dsp_chain_create would decide about the program -- validation (every constant-only DSPFatal included), LDS packing, the kernel it
 * would run on -- WITHOUT a device: no HIP call is made, so it also works on a machine that has none (the build container, a sanitizer
 * build of the planner: tests/test_planner_fuzz.py).  Same return codes and dsp_last_error() texts as dsp_chain_create.  The reference has
 * no counterpart (ProcessorManager.__init__ validates while it allocates, processing_chain.py:1527-1775). */
typedef struct dsp_plan_info {
    char kernel[64];             /* dsp_chain_kernel_name of the chain that would be created */
    char note[256];              /* dsp_chain_kernel_note */
    int32_t lds_bytes_per_wave;  /* of the generic interpreter's layout (what dsp_chain_geometry reports for it) */
    int32_t waves_per_block;
    int32_t team;                /* wavefronts per row on the interpreter: 1 or 2 */
    int32_t n_device_ops;        /* ops of the device program: the caller's, plus region-clearing ops, minus folded ones */
    int32_t lds_elems_per_wave, sreg_off, scratch_off; /* layout of a wavefront's LDS in elements of the compute type */
    int32_t n_slots;
    int32_t slot_base[DSP_MAX_SLOTS], slot_elems[DSP_MAX_SLOTS];   /* region of slot s: [base, base + elems) incl. guard and tail */
    int32_t slot_first_op[DSP_MAX_SLOTS], slot_last_op[DSP_MAX_SLOTS]; /* lifetime in ops of the caller's program */
    int32_t slot_off[DSP_MAX_SLOTS], slot_pitch[DSP_MAX_SLOTS], slot_chunk[DSP_MAX_SLOTS]; /* sample 0, lane pitch, samples per lane */
} dsp_plan_info;
int dsp_chain_plan(const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots, int n_sregs,
                   int compute_dtype, dsp_plan_info* info);
/* Enqueue one pass over n_wf rows on `stream` (asynchronous).  io_ptrs[k] = device pointer for binding k. */
int dsp_chain_execute(dsp_chain* chain, void* const* io_ptrs, int64_t n_wf, void* stream);
/* Wait for the chain's stream work and return 0 or the first data-dependent DSPFatal (code > 0) with the
 * offending row in *row (the reference raises it with wf_range, processing_chain.py:1154-1159). */
int dsp_chain_check(dsp_chain* chain, void* stream, int64_t* row);
int dsp_chain_destroy(dsp_chain* chain);
/* launch geometry chosen for the chain (for DESIGN.md / profiling): bytes of LDS per wavefront, wavefronts per
 * workgroup, workgroups for a batch of n_wf rows */
int dsp_chain_geometry(dsp_chain* chain, int64_t n_wf, int* lds_bytes_per_wave, int* waves_per_block, int* blocks);
/* Where the time goes inside the one kernel a chain is: with profiling on, the first wavefront of every workgroup times each op of
 * its waveforms with the shader clock (s_memtime).  dsp_chain_profile(chain, 1) zeroes the counters and switches it on, 0 off (the
 * fused energy kernels have no ops to time: force the interpreter with dsp_chain_set_fused(chain, 0)).  dsp_chain_profile_read waits
 * for the device and returns, per op of the device program (the caller's ops plus the region-clearing ops dsp_chain_create inserted,
 * opcode 100), its opcode, the waveform slot it reads and the summed cycles, and the number of waveforms sampled. */
int dsp_chain_profile(dsp_chain* chain, int enable);
int dsp_chain_profile_read(dsp_chain* chain, int capacity, int32_t* opcodes, int32_t* slots, uint64_t* cycles, int* n_ops,
                           uint64_t* n_waveforms);
/* name of the device kernel the chain launches (what rocprofv3 --kernel-trace lists) */
const char* dsp_chain_kernel_name(dsp_chain* chain);
/* "" or, for a chain that runs on the generic interpreter although its ops are those of a specialised kernel, the reason in a sentence (a time
 * constant per event, a length or an alignment the kernel does not take, a kernel of fewer than 64 taps ...): what a recipe's author needs to
 * see to know why a chain is 2 - 4 x slower than its neighbour.  The string lives as long as the chain. */
const char* dsp_chain_kernel_note(dsp_chain* chain);
/* Two chains of one recipe, run one behind the other on the same rows: `producer` writes pole-zero corrected rows ([bl_subtract ->] pole_zero
 * -> rows, the dsp_pz_rows_kernel shape), `consumer` is a float16 matrix-core FIR over float32 rows.  After this call the producer leaves,
 * with the rows, the per-row scale and flags the FIR would otherwise read every row once more to find; dsp_chain_execute of the consumer uses
 * them when its input is exactly what the producer's last execute wrote (address, stride, length, row count) AND it is executed on the stream
 * that execute was queued on (only stream order puts the scales ahead of the FIR), and finds them itself otherwise.  The caller's part of the
 * contract: nothing rewrites those rows between the producer's execute and the consumer's (the library cannot see a write to caller memory);
 * the note is good for one execute of the consumer.
 * A second form of the pair: `consumer` is a float16 FIR over a slice of the INTEGER rows the producer READS, minus the same per-event baseline
 * column (the cusp filter of the Ge recipes on waveform[0:6092] - baseline beside bl_subtract -> pole_zero of the whole waveform): the producer
 * sees waveform - baseline of every sample anyway and leaves the slice's scale and flags; the consumer uses them when its rows and its baseline
 * are the buffers of the producer's last execute (addresses, stride, row count, stream), and finds them itself otherwise.  A producer takes one
 * consumer of each form, a consumer one producer.
 * Returns 1 when the pair was linked, 0 when the chains are not of these shapes (nothing changes), < 0 on an argument error.  The link ends
 * with either chain's dsp_chain_destroy.  No counterpart in the reference (its processors exchange nothing but their arrays,
 * processing_chain.py:1144-1163); results are the same with and without it. */
int dsp_chain_share_row_scales(dsp_chain* producer, dsp_chain* consumer);
/* A chain of the shape LOAD [-> BL_SUBTRACT] -> POLE_ZERO -> TRAP_PICKOFF -> STORE_SCALAR (the Ge energy chain) runs on a
 * specialised kernel with the same per-sample arithmetic (dsp_energy.hip).  enable = 0 forces the generic interpreter (parity
 * tests); 1 = default specialised kernel (one chain per lane); 3 = 2 interleaved sub-chains per lane on a pad-free LDS image;
 * 5 = 4 sub-chains per lane.  Returns 1 if a specialised kernel will be used.  Environment DSPEED_HIP_NO_FUSED=1 sets the default
 * to the interpreter; DSPEED_HIP_VARIANT={1,0,2} picks the kernel variant (tuning). */
int dsp_chain_set_fused(dsp_chain* chain, int enable);
/* enable = 1: every dsp_chain_execute also copies the chain's error word to page-locked host memory on its stream, and dsp_chain_check reads
 * it there after the stream synchronisation instead of issuing a transfer of its own.  For pipelines that send the next buffer to the
 * device while the current one is processed (the reference's build_dsp.py:399-432 loop, overlapped): a transfer issued by the check
 * would queue up behind that buffer's megabytes.  One launch per chain in flight at a time, as before. */
int dsp_chain_set_async_check(dsp_chain* chain, int enable);

/* ---- linear_slope_fit over whole batches, one waveform per lane (linear_slope_fit.py:11-91) --------------------------------
 * The fit's float32 Welford recurrences are sequential per waveform, so inside a chain (one wavefront per waveform) they cost a third
 * of a LEGEND recipe's time; run over the rows of a batch with 64 waveforms per wavefront they cost nothing to speak of.  One pass
 * does up to DSP_FIT_MAX fits on windows of the waveform as the recipes read it: after an optional per-row subtraction (sub_mode 1 =
 * bl_subtract.py:11-46 with its NaN rule, 2 = numpy.subtract; sub_dev a device column of sub_dtype or NULL = sub_const) -- stage 0 --
 * and after pole_zero (pole_zero.py:24-77, has_pz, constant pz_tau in samples) on that -- stage 1.  out: n_fits x 4 columns of n_wf
 * values of the compute type (mean, stdev, slope, intercept of fit 0, then fit 1 ...).  dspeed_amd's recipe builder moves eligible
 * fits of a recipe here and feeds the columns to the chain as per-event inputs. */
#define DSP_FIT_MAX 4
typedef struct dsp_fit_window {
    int32_t stage; /* 0: the (subtracted) waveform, 1: its pole-zero correction */
    int32_t first; /* window [first, first + count) in samples of the row as bound (wf .. wf + wf_len) */
    int32_t count;
} dsp_fit_window;
int dsp_linear_slope_fit_rows(const void* wf, int wf_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, int compute_dtype,
                              const void* sub_dev, int sub_dtype, double sub_const, int sub_mode, int has_pz, double pz_tau,
                              const dsp_fit_window* fits, int n_fits, void* out, void* stream);

/* ---- single processors: the gufunc entry points -----------------------------------------------------
 * One call = one reference gufunc call on an (n_wf, wf_len) block: `in`/`out` device pointers, rows
 * `*_stride` elements apart.  Scalar gufunc arguments "()" come as (pointer, value): if the pointer is
 * non-NULL it is a device array with one value per waveform, otherwise `value` is broadcast.
 * The call is synchronous with respect to errors: it returns DSP_OK or the DSP_E_* code (first offending
 * row in *err_row when not NULL).  `<ty>` names the gufunc loop: `_f32` takes float32, int16 or uint16 rows (in_dtype) and
 * produces float32; `_f64` takes float64, int32 or uint32 rows and produces float64 -- the reference's type matching
 * (first signature every argument can be cast to, processing_chain.py:1565-1572, 1654-1664).
 */
int dsp_bl_subtract_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* baseline_dev,
                        float baseline, float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_pole_zero_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float tau, float* out,
                      int64_t out_stride, void* stream, int64_t* err_row);
int dsp_double_pole_zero_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float tau1, float tau2,
                             float frac, float* out, int64_t out_stride, void* stream, int64_t* err_row);
/* pole_zero / double_pole_zero with a time constant (fraction) per waveform: the gufunc layouts "(n),()->(n)" / "(n),(),(),()->(n)" let
 * ProcessorManager broadcast a per-event variable into a "()" slot (pole_zero.py:24-30, 82-90).  Each *_dev is a device column of n_wf
 * values or NULL for the constant beside it. */
int dsp_pole_zero_col_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* tau_dev, float tau,
                          float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_double_pole_zero_col_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* tau1_dev,
                                 float tau1, const float* tau2_dev, float tau2, const float* frac_dev, float frac, float* out,
                                 int64_t out_stride, void* stream, int64_t* err_row);
int dsp_pole_zero_col_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* tau_dev, double tau,
                          double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_double_pole_zero_col_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* tau1_dev,
                                 double tau1, const double* tau2_dev, double tau2, const double* frac_dev, double frac, double* out,
                                 int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_filter_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                        float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_norm_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                      float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_asym_trap_filter_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise,
                             int32_t flat, int32_t fall, float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_fixed_time_pickoff_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* t_in_dev,
                               float t_in, int32_t mode_char, float* out, void* stream, int64_t* err_row);
int dsp_time_point_thresh_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,
                              const float* threshold_dev, float threshold, const float* t_start_dev, float t_start,
                              float walk_forward, float* out, void* stream, int64_t* err_row);
int dsp_interpolated_time_point_thresh_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,
                                           const float* threshold_dev, float threshold, const float* t_start_dev, float t_start,
                                           int64_t walk_forward, int32_t mode_char, float* out, void* stream, int64_t* err_row);
int dsp_min_max_norm_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* a_min_dev, float a_min,
                         const float* a_max_dev, float a_max, float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_windower_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* t0_dev, float t0,
                     float* out, int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_avg_current_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float length, float* out,
                        int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_pickoff_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                         const float* t_pickoff_dev, float t_pickoff, float* out, void* stream, int64_t* err_row);
int dsp_upsampler_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float upsample, float* out,
                      int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_moving_window_multi_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float length, float num_mw,
                                int32_t mw_type, float* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_linear_slope_fit_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float* mean, float* stdev,
                             float* slope, float* intercept, void* stream, int64_t* err_row);
int dsp_mean_below_threshold_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* threshold_dev,
                                 float threshold, float* out, void* stream, int64_t* err_row);
int dsp_min_max_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, float* t_min, float* t_max,
                    float* a_min, float* a_max, void* stream, int64_t* err_row);
int dsp_dwt_haar_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t level, int32_t coeff_char,
                     float* out, int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_convolve_wf_f32(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const float* kernel_dev,
                        int32_t kernel_len, int32_t mode_char, float* out, int32_t out_len, int64_t out_stride, void* stream,
                        int64_t* err_row);

/* the float64 loops: float64 (and int32 / uint32) rows, float64 scalars and outputs -- same argument order */
int dsp_bl_subtract_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* baseline_dev,
                        double baseline, double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_pole_zero_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double tau, double* out,
                      int64_t out_stride, void* stream, int64_t* err_row);
int dsp_double_pole_zero_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double tau1, double tau2,
                             double frac, double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_filter_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                        double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_norm_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                      double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_asym_trap_filter_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise,
                             int32_t flat, int32_t fall, double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_fixed_time_pickoff_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* t_in_dev,
                               double t_in, int32_t mode_char, double* out, void* stream, int64_t* err_row);
int dsp_time_point_thresh_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,
                              const double* threshold_dev, double threshold, const double* t_start_dev, double t_start,
                              double walk_forward, double* out, void* stream, int64_t* err_row);
int dsp_interpolated_time_point_thresh_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,
                                           const double* threshold_dev, double threshold, const double* t_start_dev, double t_start,
                                           int64_t walk_forward, int32_t mode_char, double* out, void* stream, int64_t* err_row);
int dsp_min_max_norm_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* a_min_dev, double a_min,
                         const double* a_max_dev, double a_max, double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_windower_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* t0_dev, double t0,
                     double* out, int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_avg_current_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double length, double* out,
                        int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_trap_pickoff_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t rise, int32_t flat,
                         const double* t_pickoff_dev, double t_pickoff, double* out, void* stream, int64_t* err_row);
int dsp_upsampler_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double upsample, double* out,
                      int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_moving_window_multi_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double length, double num_mw,
                                int32_t mw_type, double* out, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_linear_slope_fit_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double* mean, double* stdev,
                             double* slope, double* intercept, void* stream, int64_t* err_row);
int dsp_mean_below_threshold_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride,
                                 const double* threshold_dev, double threshold, double* out, void* stream, int64_t* err_row);
int dsp_min_max_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, double* t_min, double* t_max,
                    double* a_min, double* a_max, void* stream, int64_t* err_row);
int dsp_dwt_haar_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, int32_t level, int32_t coeff_char,
                     double* out, int32_t out_len, int64_t out_stride, void* stream, int64_t* err_row);
int dsp_convolve_wf_f64(const void* in, int in_dtype, int64_t n_wf, int32_t wf_len, int64_t in_stride, const double* kernel_dev,
                        int32_t kernel_len, int32_t mode_char, double* out, int32_t out_len, int64_t out_stride, void* stream,
                        int64_t* err_row);

/* ---- synthetic batches generated on the device (bench.py) ---------------------------------
 * wf[r][i] = B_r + A_r*exp(-(i-t0_r)/tau)*[i>=t0_r] + sigma*n(r,i), counter-based hash noise;  also writes the
 * per-waveform baseline B_r and the pick-off time t0_r + pick_offset (samples).  out_dtype DSP_F32 or DSP_I16. */
int dsp_synth_waveforms(void* wf, int out_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, float* baseline, float* t_pick,
                        uint64_t seed, int64_t first_row, float tau, float sigma, float pick_offset, float bl_lo, float bl_hi,
                        float amp_lo, float amp_hi, void* stream);

/* The same rows with a charge-collection time: the step reaches its height linearly over `rise` samples, rise uniform in
 * [rise_lo, rise_hi] per waveform (the recipe benchmarks: the rise-time walks of a Ge recipe end a few samples from their start on
 * pulses that take tens of samples to rise, as the detectors' do, and walk the whole waveform on a one-sample step). */
int dsp_synth_pulses(void* wf, int out_dtype, int64_t n_wf, int32_t wf_len, int64_t row_stride, float* baseline, float* t_pick, uint64_t seed,
                     int64_t first_row, float tau, float sigma, float pick_offset, float bl_lo, float bl_hi, float amp_lo, float amp_hi,
                     float rise_lo, float rise_hi, void* stream);

/* ---- measurement helper (bench.py; SURVEY.md 8d "also report against a measured read-only streaming kernel") -----------
 * Reads `bytes` (multiple of 16) device bytes once with 16-byte loads and discards them; `sink` = any 4 writable device bytes.
 * Asynchronous on `stream`.  Replaces nothing in the reference. */
int dsp_stream_read(const void* src, int64_t bytes, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSPEED_HIP_H */
