/*
 * mi355gate_debug.h -- development surface of libmi355gate.so: stage taps for the parity tests, A/B and
 * fault-injection options, per-kernel event timing for bench.py.  Not part of the drop-in boundary
 * (mi355gate.h): nothing here replaces a reference interface, and a caller of the product ABI never needs it.
 * Same library, same handle; the entry points are exported next to the product ones.
 */
#ifndef MI355GATE_DEBUG_H
#define MI355GATE_DEBUG_H

#include "mi355gate.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stage taps (parity tests) ---------------------------------------------------------- */
/* Fields of the last processed unit batch, copied to the host (synchronises):
 * what = 0: raw mask  float[units][T][FS];  1: final mask float[units][T][FS];
 *        2: power     double[units][T][FS] (stationary, materialised path only);
 *        3: raw mask as bits uint64[units][T][ceil(F/64)] (fused stationary path only).
 * FS = third entry of sg_debug_dims. */
SG_API int sg_debug_dims(const sg_handle* h, int64_t dims[3]); /* units, T, FS */
/* Frame range [t0, t1) for which field 3 (mask bits) was computed: the fast path only decides
 * the frames that reach the kept output samples (+- the smoothing half width). */
SG_API int sg_debug_range(const sg_handle* h, int64_t range[2]);
/* Diagnostic counters (synchronises `stream`).  which = 0: (row, band) pairs of the one-kernel TorchGate row gate that were
 * re-evaluated in float64 since the handle was created (the float32 statistics could not decide them within their error
 * bound); divide by rows x 513 for the rate.  which = 1 / 2: batches of the one-pass gate that took the in-kernel / the a-priori
 * floor test (SG_OPT_FLOOR_TEST) since the handle was created (host counters, no synchronisation); which = 3: launch epoch of the
 * last gate call in which some chunk's floor test fired (0: never; synchronises).  which = 4 .. 16: development builds of the
 * persistent one-pass gate only (-DOP_TILECOUNT / -DOP_WHO / -DOP_TRACE libraries, tools/experiments/persist_*.py: words of the
 * host-mapped error block; 8: unfinished tiles of the phase trace; 16: the per-ticket draw / start record around ticket *value,
 * on stderr) -- they return 0 or SG_E_INVALID in the product library. */
SG_API int sg_debug_counter(sg_handle* h, int32_t which, int64_t* value, void* stream);
SG_API int sg_debug_fetch(sg_handle* h, int32_t what, void* host, int64_t bytes, void* stream);

/* ---- development options (sg_set_option / sg_get_option of mi355gate.h) ----------------- */
#define SG_OPT_FORCE_UNFUSED 1 /* value != 0: use the materialised (v1) kernels everywhere */
#define SG_OPT_FORCE_F64_DECIDE 3 /* value != 0: decide every mask cell from a float64 STFT */
#define SG_OPT_FORCE_NOSEAM 4   /* value != 0: overlapping apply tiles instead of abutting tiles + seam kernel */
#define SG_OPT_FORCE_NOLEAN 5   /* value != 0: apply kernel with full-size LDS slices and stored frames */
#define SG_OPT_FORCE_SPLIT 6    /* value != 0: default geometry: decide / smooth / apply as three kernels instead of the one-pass kernel; non-stationary gate: the serial tile chain (k_iir_comb + k_iir_chain) instead of k_iir_chain_par */
#define SG_OPT_INJECT_HANDOFF_FAULT 7 /* tests: the next launch with in-launch hand-offs reports `value` (bits 0..2) as lost hand-offs
                                       * (the output is fine); bits 3..5 make the KERNEL lose its hand-offs (3 or 4: the one-pass
                                       * gate, 5: the fused apply): the producers' tags are never accepted and the polls give up --
                                       * the bounded-poll timeout path itself: error word set by the kernel, affected hops NaN */
#define SG_OPT_FORCE_NOROWGATE 10 /* variant T, stationary, rows of <= 64 frames: 0 (default) = the one-kernel row gate for calls of >= 160
                                   * rows, the four-kernel path (float64 transform of every frame, k_row_decide, k_smooth_bits2,
                                   * k_apply_fast) below; 1 = never the row gate; 2 = the row gate whenever the shape is eligible */
#define SG_OPT_ROWGATE_TAP 11     /* value != 0: the row gate also writes its float32 power tile (4 |X|^2, [rows][64][528]) for
                                   * sg_debug_fetch(what = 4): measurements behind the decision margin */
#define SG_OPT_ROWGATE_SHAPE 12   /* value = 16 (default) or 8: wavefronts per workgroup of the row gate (16 x one quad of frames at 128
                                   * VGPRs, or 8 x two quads at 256 VGPRs without scratch): A/B measurements */
#define SG_OPT_FLOOR_TEST 13      /* one-pass gate (k_gate_onepass): how "can _amp_to_db's -top_db floor lift a band of this chunk over its
                                   * threshold?" is answered.  1 = a priori (k_unit_absmax reads the recording once more before the gate);
                                   * 2 = by the gate kernel on the samples it stages -- free unless a chunk reports, which is then gated a
                                   * second time with its float64 band maxima; 0 (default) = predicted from what recent calls on the handle
                                   * found (no synchronisation).  Same result either way: exact band maxima decide */
#define SG_OPT_TILE_ORDER 15      /* one-pass gate (n_fft = 1024): how workgroups come by their tiles.  0 (default, round 6) = PERSISTENT
                                   * workgroups: three per compute unit, each looping over atomic tickets -- constant tables staged once, the
                                   * next ticket drawn and the next tile's samples prefetched under the current tile's second half (1 % faster
                                   * than 2 for one call at a time; bit-identical).  2 = one ticket-drawn tile per workgroup (rounds 2-5: the better form when SEVERAL
                                   * calls are in flight on their own handles and streams -- persistent workgroups hold every workgroup slot until their launch
                                   * ends: 111 against 97 Gsamples/s at two calls in flight, bench.py --streams; REDO launches, a-priori floor
                                   * flags and the fault-injection instantiation always run this form).
                                   * 1 = tile = block index -- no atomic at all, at the price of assuming that the dispatcher starts workgroups
                                   * in index order (it does on gfx950; HIP does not promise it).  Waits stay bounded and reported in every
                                   * mode; on an otherwise idle GPU the outputs are bit-identical */
#define SG_OPT_EXACT_MATERIALISED 16 /* value != 0: the float64 pipeline (integer outputs, SG_OPT_FORCE_EXACT) as it was before round 5 -- every
                                      * field materialised, the recurrence serial per band, two direct smoothing passes, masked frames + gather --
                                      * also where the fused float64 apply (k_apply_fast64), the tile-parallel recurrence and the LDS-tiled
                                      * smoothing would do: A/B and parity of the two */
#define SG_OPT_FORCE_NOFAST 2  /* value != 0: keep the bit-mask stages but use the general apply kernels */

/* ---- per-kernel timing (bench.py's roofline leg) ------------------------------------- */
#define SG_STAGE_CHANNEL_MEAN 0
#define SG_STAGE_STFT_POWER 1
#define SG_STAGE_COLMAX 2
#define SG_STAGE_COLSTATS 3
#define SG_STAGE_DECIDE 4
#define SG_STAGE_STFT_MAG 5
#define SG_STAGE_NONSTAT_MASK 6
#define SG_STAGE_SMOOTH 7
#define SG_STAGE_APPLY_ISTFT 8
#define SG_STAGE_OLA 9
#define SG_STAGE_NOISE_STATS 10 /* every launch of sg_noise_stats */
#define SG_STAGE_PREP 11
#define SG_STAGE_STFT_MAX 12
#define SG_STAGE_STFT_BITS 13
#define SG_STAGE_APPLY_FAST 14
#define SG_STAGE_DECIDE_FAST 15
#define SG_STAGE_ONEPASS 16
#define SG_STAGE_ROW_GATE 17   /* k_row_gate: TorchGate.forward of a whole row (<= 64 frames) in one kernel */
#define SG_STAGE_IIR_CHAIN 18 /* non-stationary gate: carries of the time tiles (k_iir_chain_par; serial form k_iir_part / k_iir_comb + k_iir_chain) */
#define SG_STAGE_IIR_MASK 19  /* non-stationary gate: k_iir_mask<nt> -- recurrence, sigmoid and both smoothing passes in one kernel */
#define SG_N_STAGES 20
/* When enabled, every kernel launch is bracketed by a hipEvent pair recorded on the launch
 * stream.  sg_profile_read synchronises those events and returns accumulated milliseconds
 * and launch counts per stage (arrays of SG_N_STAGES); reset != 0 clears the accumulators. */
SG_API int sg_profile_enable(sg_handle* h, int32_t on);
/* Restrict the event pairs to the stages whose bit (1 << SG_STAGE_*) is set; 0 = all stages.  Timing
 * one kernel costs two event records per step instead of ~30. */
SG_API int sg_profile_select(sg_handle* h, int64_t stage_mask);
SG_API int sg_profile_read(sg_handle* h, double* ms, int64_t* counts, int32_t n_stages, int32_t reset);
SG_API const char* sg_stage_name(int32_t stage);

#ifdef __cplusplus
}
#endif
#endif /* MI355GATE_DEBUG_H */
