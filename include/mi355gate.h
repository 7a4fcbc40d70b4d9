/*
 * mi355gate.h -- C ABI of libmi355gate.so, the MI355X (gfx950) spectral-gating engine.
 *
 * This is the drop-in boundary for ONE hot path of timsainb/noisereduce: the chunk
 * filter  STFT -> per-band noise statistics -> threshold/sigmoid mask -> 2-D mask
 * smoothing -> masked complex multiply -> overlap-add ISTFT.  Nothing like this ABI
 * exists in the reference (it is pure Python); each entry point states which reference
 * interface it replaces (paths relative to /root/reference/noisereduce/).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All `*_dev` pointers are DEVICE
 *     pointers (e.g. torch.Tensor.data_ptr() of a ROCm tensor); `*_host` are host pointers.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Every call only
 *     ENQUEUES work on that stream unless documented as synchronising.
 *   - return value: 0 = success, negative = error (SG_E_*); text via sg_last_error().
 *     Nothing throws across the boundary.
 *   - a handle is not thread-safe; use one handle per (device, stream).
 *   - caller owns input/output buffers; the library owns its workspace (grown on demand,
 *     bounded by sg_params.max_workspace_bytes; large jobs are processed in unit batches).
 *   - non-finite samples: a NaN behaves as in the reference (spectralgate/stationary.py:75-106,
 *     torchgate/torchgate.py:140-160: numpy / torch maxima and means keep it): every band of the
 *     chunk / row that sees it is gated, a NaN in the noise clip gates everything, the NaN itself
 *     survives in the output.  An Inf sample is treated like a NaN.
 */
#ifndef MI355GATE_H
#define MI355GATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_VERSION 100 /* 0.1.0 */

/* The library is built with -fvisibility=hidden: only the entry points declared here are exported. */
#ifndef SG_API
#define SG_API __attribute__((visibility("default")))
#endif

/* error codes */
#define SG_OK 0
#define SG_E_INVALID (-1)     /* bad argument (maps to ValueError)            */
#define SG_E_UNSUPPORTED (-2) /* geometry outside the kernels' range          */
#define SG_E_HIP (-3)         /* HIP runtime error                            */
#define SG_E_NOMEM (-4)       /* workspace allocation failed                  */
#define SG_E_STATE (-5)       /* call order (e.g. no noise threshold yet)     */
#define SG_E_HANDOFF (-6)     /* a bounded inter-workgroup wait timed out: output invalid, re-run (sg_check_errors) */

/* sample dtypes of caller buffers */
#define SG_F32 0
#define SG_F64 1
#define SG_I16 2
#define SG_I32 3

/* algorithm variant: the reference holds two different algorithms behind one API
 *  */
#define SG_VARIANT_S 0 /* numpy/scipy "spectralgate": spectralgate/{base,stationary,nonstationary}.py */
#define SG_VARIANT_T 1 /* torch "torchgate": torchgate/torchgate.py                                   */

typedef struct sg_params {
  int32_t variant;    /* SG_VARIANT_S | SG_VARIANT_T                                        */
  int32_t stationary; /* 1: stationary (threshold) mask, 0: non-stationary (sigmoid) mask   */
  int32_t n_fft;      /* 4..32768, or a power of two up to 65536 (base.py:77; torchgate.py:55) */
  int32_t win_length; /* <= n_fft (base.py:79-82)                                           */
  int32_t hop_length; /* >= 1    (base.py:83-86)                                            */
  int32_t n_grad_freq; /* mask-smoothing half width in bins  (base.py:104), >= 1            */
  int32_t n_grad_time; /* mask-smoothing half width in frames (base.py:115), >= 1           */
  int32_t smooth_mask; /* 0: no smoothing (base.py:90-91,124-125)                           */
  int64_t chunk_size;  /* S only: base.py:66 (reduce_noise default 600000)                  */
  int64_t padding;     /* S only: base.py:67 (default 30000)                                */
  double prop_decrease;     /* base.py:88; torchgate.py:53                                  */
  double n_std_thresh;      /* stationary.py:45,79-81; torchgate.py:160                     */
  double top_db;            /* 80 (spectralgate/utils.py:11) or 40 (torchgate/utils.py:6)   */
  int32_t ddof;             /* 0: np.std (stationary.py:77); 1: torch.std_mean (torchgate.py:158) */
  int32_t n_movemean;       /* T non-stationary: boxcar length (torchgate.py:179-190)       */
  double nonstat_thresh;    /* S: thresh_n_mult_nonstationary; T: n_thresh_nonstationary     */
  double nonstat_slope;     /* S: sigmoid_slope_nonstationary; T: 1 / temp_coeff_nonstationary */
  double iir_b;             /* S non-stationary: one-pole coefficient (nonstationary.py:109-114) */
  int64_t max_workspace_bytes; /* 0 = default (8 GiB) */
} sg_params;

typedef struct sg_handle sg_handle;

SG_API int sg_version(void);

/* Last error text of a handle (or of the last failed sg_create when h == NULL). */
SG_API const char* sg_last_error(const sg_handle* h);

/* Replaces SpectralGate.__init__ parameter resolution (base.py:33-97) and
 * TorchGate.__init__ (torchgate.py:31-71): builds twiddle/window/smoothing tables on the
 * current HIP device.  `window_host`: win_length doubles, or NULL for the periodic Hann
 * window both references use (scipy get_window('hann'), torch.hann_window). */
SG_API int sg_create(const sg_params* p, const double* window_host, sg_handle** out);
SG_API int sg_destroy(sg_handle* h);

/* Geometry helpers: number of STFT frames and ISTFT output length for a length-L signal
 * (scipy/_spectral_py.py:2185-2189,1715; torch.stft/istft center=True). */
SG_API int sg_n_frames(const sg_handle* h, int64_t L, int64_t* n_frames);
SG_API int sg_output_length(const sg_handle* h, int64_t L, int64_t* out_len);

/* Workspace (bytes of HBM) the handle will own after an sg_process_chunks call on a (C, N) recording
 * (chunked as in sg_process_chunks) -- so that a caller can budget device memory next to the recording
 * (base.py:180-216 streams through a memmap instead; here units are processed in batches bounded by
 * sg_params.max_workspace_bytes).  An upper bound: it includes the exchange buffers of the fused kernels and the
 * float32 copy that a recording of another sample type (int16 / int32 / float64) costs on the default geometry --
 * C * N * 4 bytes, which a float32 recording does not pay.  The entry point does not know the sample type of the call
 * to come: unless SG_OPT_FAST_INTEGER is set (and SG_OPT_FORCE_EXACT clear) it reports the LARGER of the fused float32
 * pipeline and the float64 pipeline that integer outputs take by default (32 bytes per time-frequency cell plus float64
 * frames, batched under max_workspace_bytes / the 8 GiB default); with SG_OPT_FORCE_EXACT, the float64 figure.
 * Pure host arithmetic, no device work. */
SG_API int sg_workspace_bytes(const sg_handle* h, int64_t C, int64_t N, int32_t chunked, int64_t* bytes);

/* ---- variant S -------------------------------------------------------------------- */

/* Replaces the noise-statistics block of SpectralGateStationary.__init__
 * (stationary.py:47-81): channel mean of the (C, n) noise clip, STFT, dB with -top_db
 * floor, per-band mean/std over time, thresh = mean + n_std*std.  The caller applies
 * clip_noise_stationary (n = min(n, chunk_size)).  Result stays on the device. */
SG_API int sg_noise_stats(sg_handle* h, const void* noise_dev, int dtype, int64_t C, int64_t n,
                   int64_t row_stride, void* stream);
/* Read back / override the per-band threshold in dB (n_fft/2+1 doubles).
 * sg_get_noise_threshold synchronises the stream. */
SG_API int sg_get_noise_threshold(sg_handle* h, double* thresh_host, int32_t n_bins, void* stream);
SG_API int sg_set_noise_threshold(sg_handle* h, const double* thresh_host, int32_t n_bins, void* stream);
/* Device-to-device forms (asynchronous on `stream`, no host synchronisation): used to broadcast
 * the threshold between ranks with RCCL. */
SG_API int sg_get_noise_threshold_dev(sg_handle* h, double* thresh_dev, int32_t n_bins, void* stream);
SG_API int sg_set_noise_threshold_dev(sg_handle* h, const double* thresh_dev, int32_t n_bins, void* stream);

/* Replaces SpectralGate.get_traces + filter_chunk + _read_chunk + _do_filter for a whole
 * (C, N) planar recording that already lives in HBM (base.py:130-226): the reference's
 * chunk grid (chunk i = samples [i*cs, (i+1)*cs), filtered on a zero-padded window of
 * `padding` extra samples per side, padding discarded) is evaluated on the device, all
 * (channel, chunk) units in one set of launches.  Writes out[c][g - start_frame] for
 * g in [start_frame, end_frame); pass 0, N for everything.  `chunked` = 0 reproduces the
 * single-window branch (base.py:222), 1 the chunk grid (base.py:175-216).
 * `halo_left` / `halo_right`: number of valid samples stored BEFORE index 0 / AFTER index N-1
 * of every row (0 for a plain recording).  A rank that holds one time shard of a longer
 * recording passes its neighbours' seam samples this way so that chunk windows read real
 * data instead of zeros across the shard boundary. */
SG_API int sg_process_chunks(sg_handle* h, const void* in_dev, int in_dtype, void* out_dev,
                      int out_dtype, int64_t C, int64_t N, int64_t in_stride,
                      int64_t out_stride, int64_t start_frame, int64_t end_frame,
                      int32_t chunked, int64_t halo_left, int64_t halo_right, void* stream);

/* Replaces SpectralGate._do_filter(chunk) (base.py:158-160; stationary.py:129-133;
 * nonstationary.py:99-103): (C, Lp) padded chunk in, (C, Lp) filtered chunk out, the
 * last Lp - sg_output_length(Lp) samples are zero like the reference's. */
SG_API int sg_filter_padded(sg_handle* h, const void* chunk_dev, int in_dtype, void* out_dev,
                     int out_dtype, int64_t C, int64_t Lp, int64_t in_stride,
                     int64_t out_stride, void* stream);

/* ---- variant T -------------------------------------------------------------------- */

/* Replaces TorchGate.forward(x, xn) (torchgate.py:200-264): x (B, L) -> out
 * (B, sg_output_length(L)) = hop*(T-1) (+1 for an odd n_fft), T = sg_n_frames(L) =
 * 1 + (L + 2*(n_fft/2) - n_fft)/hop -- hop*(L/hop) for an even n_fft.  xn_dev may be NULL (statistics from x itself, per row) or a
 * (Bn, Ln) noise batch with Bn in {1, B}.
 * mask_out_dev: NULL, or a float[B][T][FS] buffer (T = sg_n_frames(L), FS = round_up(n_fft/2+1,16))
 * that receives the final (smoothed) mask, for sg_process_batch_backward. */
SG_API int sg_process_batch(sg_handle* h, const void* x_dev, int dtype, int64_t B, int64_t L,
                     int64_t x_stride, const void* xn_dev, int64_t Bn, int64_t Ln,
                     int64_t xn_stride, void* out_dev, int out_dtype, int64_t out_stride,
                     float* mask_out_dev, void* stream);

/* Adjoint of sg_process_batch with the mask held fixed (TorchGate.forward is differentiable
 * w.r.t. x with the mask detached, torchgate.py:126,167; the reference gets this from autograd
 * through torch.stft/istft): grad_out (B, Lout) -> grad_x (B, L), both of sample type `dtype`
 * (SG_F32 or SG_F64).  mask_dev = the buffer filled by sg_process_batch(mask_out_dev). */
SG_API int sg_process_batch_backward(sg_handle* h, const void* grad_out_dev, int dtype, int64_t B,
                              int64_t L, int64_t go_stride, const float* mask_dev,
                              void* grad_x_dev, int64_t gx_stride, void* stream);

/* ---- stage taps (used by the parity tests; also plain STFT/ISTFT operators) ------ */

/* Forward STFT of (B, L) rows -> complex float64 Z[B][T][F] (interleaved re,im), same
 * scaling as the variant's reference call (S: 1/sum(w), scipy stft; T: unscaled). */
SG_API int sg_stft(sg_handle* h, const void* x_dev, int dtype, int64_t B, int64_t L, int64_t stride,
            double* z_dev, void* stream);
/* ---- options ------------------------------------------------------------------------- */
/* Product options.  The A/B, fault-injection and profiling switches the tests and tools use live in
 * mi355gate_debug.h (same library, same sg_set_option). */
#define SG_OPT_FAST_INTEGER 8   /* value != 0: integer (SG_I16 / SG_I32) outputs from the fused float32 kernels: <= 1 LSB away from
                                 * the reference on ~1 % of the samples.  Default: the float64 pipeline, whose truncated result IS
                                 * the reference's (base.py:217-226 casts a float64 array), an order of magnitude slower */
#define SG_OPT_FORCE_EXACT 9    /* value != 0: float64 pipeline for every output dtype (float64 recordings: float64-accurate results) */
SG_API int sg_set_option(sg_handle* h, int32_t option, int64_t value);
/* Current value of an option (the library's default if it was never set). */
SG_API int sg_get_option(const sg_handle* h, int32_t option, int64_t* value);

/* ---- deferred device-side errors -------------------------------------------------------- */
/* The fused kernels of the default geometry hand data from workgroup to workgroup INSIDE a launch (mask bits,
 * partial hops; placement-independent ticket protocol, every wait bounded to about a second).  A wait that
 * times out -- the device was preempted or time-sliced for that long -- cannot be reported by the asynchronous
 * call that enqueued the kernel.  sg_check_errors synchronises `stream` and returns SG_E_HANDOFF when a launch
 * enqueued on this handle since the previous check lost a hand-off: those calls' outputs are invalid and must
 * be re-run.  Entry points that synchronise anyway (sg_get_noise_threshold; the stage taps of mi355gate_debug.h) report
 * the same way; a caller
 * that never checks gets SG_E_HANDOFF from its NEXT compute call on the handle -- and never plausible-looking audio:
 * a tile that lost a hand-off writes NaN to every output hop it could not finalise.  The Python layer checks after
 * every call that returns host arrays and re-runs a failed call on the kernels without in-launch hand-offs.
 * (No counterpart in the reference: base.py:206-216 joins its joblib workers.) */
SG_API int sg_check_errors(sg_handle* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355GATE_H */
