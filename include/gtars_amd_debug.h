/*
 * gtars_amd_debug.h -- test, A/B and diagnostics hooks of libgtars_amd.so.  NOT part of the drop-in boundary
 * (include/gtars_amd.h, include/gtars_amd_host.h): nothing a binding of the reference's API needs is declared here, and
 * these entry points may change with the kernels they look into.  Same status / error conventions as gtars_amd.h.
 */
#ifndef GTARS_AMD_DEBUG_H
#define GTARS_AMD_DEBUG_H

#include "gtars_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test / diagnostics hook: launches `workgroups` workgroups of 1024 threads that each hold `lds_bytes` of LDS and spin
 * for `microseconds` on `stream` -- a stand-in for foreign work that occupies CUs while a tokenizer launch runs on
 * another stream (tests/test_gpu_parity.py: the chained scan must complete with the right result whatever is resident). */
gtars_status gtars_debug_occupy_device(void *stream, uint32_t workgroups, uint32_t lds_bytes, uint32_t microseconds);

/* Test / A-B hook.  The library reads its GTARS_* environment switches (test and ablation knobs: GTARS_IGD_SWEEP_MIN,
 * GTARS_NO_LDS_PATH, GTARS_HOST_THREADS ...) ONCE, into an immutable snapshot taken at first use -- never with a getenv per
 * call, which races with a host program's setenv.  A process that changes a switch afterwards calls this to make the library take
 * a new snapshot; no other library call may be in flight. */
void gtars_debug_reload_env(void);

/* Test hook for the handles' device affinity (gtars_index_device): overwrite the device id a handle records -- of an index
 * (and its flat companion) when is_igd == 0, of an IGD (and its pieces view) otherwise -- and return the previous one.  The
 * handle's memory does not move: a forged id makes `*_device` entry points refuse the call and host-buffer entry points try
 * to switch to that device, which is what tests/test_gpu_parity.py asserts on a one-GPU box.  Put the real id back before
 * any other use. */
int gtars_debug_set_handle_device(void *handle, int is_igd, int device);

/* Test / measurement entry of the device-side DEFLATE decoder (csrc/inflate_dev.hip; the fused fragment pipeline's input stage,
 * gtars-fragsplit/src/split.rs:84-131 reads its files through flate2's MultiGzDecoder): n_streams raw DEFLATE streams (RFC 1951, no
 * gzip header or trailer), one wave each.  ALL pointers are device memory.  Stream i is comp[in_off[i] .. + in_len[i]) -- in_off a
 * multiple of 16, and 32 readable bytes behind every stream -- and is written to out[out_off[i] ..) -- out_off a multiple of 16,
 * capacity out_cap[i] bytes.  out_len[i] = bytes produced, consumed[i] = input bytes used, status[i] = 0 or the reason the decoder
 * refused the stream (1 block type / stored length, 2 code lengths, 3 symbol or distance, 4 input exhausted, 5 capacity).
 * Enqueued on `stream`; returns a gtars_status. */
int gtars_debug_inflate_streams(const void *comp, const uint64_t *in_off, const uint32_t *in_len, void *out, const uint64_t *out_off,
                                const uint32_t *out_cap, uint32_t n_streams, uint32_t *out_len, uint32_t *consumed, uint32_t *status,
                                void *stream);

/* (Stamp builds -- tools/build_variant.sh with -DGTARS_TOK_STAMPS=1 / -DIGD_STAMPS=1 -- additionally export
 * gtars_debug_tok_stamps / gtars_debug_route_stamps / gtars_debug_sweep_stamps: s_memtime at the phase boundaries of the tile
 * loops, read by tools/r03_tok_stamps.py, tools/r03_sweep_stamps.py, tools/r05_rank_stamps.py.  The shipped library has none.) */

#ifdef __cplusplus
}
#endif
#endif
