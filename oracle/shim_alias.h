/*
 * shim_alias.h -- TEST INFRASTRUCTURE ONLY (build container; never shipped, never on the product path).
 *
 * integration/hipshim.c is the Linrad-side glue of liblinrad_hip.so.  To EXECUTE it without a GPU it is compiled a second time
 * with this header force-included (-include): every context-taking lrh_* call resolves to the oracle's lro_* function of the same
 * shape (oracle/linrad_oracle.h mirrors include/linrad_hip.h), so the patched reference objects + hipshim.c + liblinrad_oracle.so
 * link into oracle/_ref/shim_harness and run the reference's own call sites end to end (oracle/build_shim_harness.sh,
 * tests/test_shim_exec_cpu.py).  What has no oracle counterpart is host plumbing of the GPU library (page-locking, the
 * asynchronous copy stream) and is stubbed here: the copy becomes the synchronous lro_timf1_write.
 * lrh_config_defaults (no context, plain host code) comes from liblinrad_hip.so itself.
 */
#ifndef SHIM_ALIAS_H
#define SHIM_ALIAS_H
#define lrh_ctx lro_ctx
#include "../include/linrad_hip.h"
#include "linrad_oracle.h"

#define lrh_open lro_open
#define lrh_close lro_close
#define lrh_export lro_export
#define lrh_export_fft1_net lro_export_fft1_net
#define lrh_fft1_b lro_fft1_b
#define lrh_fft1_c lro_fft1_c
#define lrh_make_timf2 lro_make_timf2
#define lrh_first_noise_blanker lro_first_noise_blanker
#define lrh_make_fft2 lro_make_fft2
#define lrh_fft2_mix1_fixed lro_fft2_mix1_fixed
#define lrh_fft1_mix1_fixed lro_fft1_mix1_fixed
#define lrh_fft2_mix1_afc lro_fft2_mix1_afc
#define lrh_fft1_mix1_afc lro_fft1_mix1_afc
#define lrh_compute_timf2_powersum lro_compute_timf2_powersum
#define lrh_fft1_update_liminfo lro_fft1_update_liminfo
#define lrh_fft2_update_liminfo lro_fft2_update_liminfo
#define lrh_get_blanker_state lro_get_blanker_state
#define lrh_get_liminfo lro_get_liminfo
#define lrh_get_liminfo_amplitude_factor lro_get_liminfo_amplitude_factor
#define lrh_get_mix1_state lro_get_mix1_state
#define lrh_set_blanker_tables lro_set_blanker_tables
#define lrh_spur_config lro_spur_config
#define lrh_spur_acquire lro_spur_acquire
#define lrh_spur_get lro_spur_get
#define lrh_spur_permute lro_spur_permute
#define lrh_spur_search_config lro_spur_search_config
#define lrh_spur_search_get lro_spur_search_get
#define lrh_blanker_begin lro_blanker_begin
#define lrh_blanker_weak_span lro_blanker_weak_span
#define lrh_blanker_finish lro_blanker_finish
#define lrh_exchange_read lro_exchange_read
#define lrh_exchange_write lro_exchange_write
#define lrh_fft2_xy_begin lro_fft2_xy_begin
#define lrh_fft2_xy_finish lro_fft2_xy_finish
#define lrh_set_ch2_phasing lro_set_ch2_phasing
#define lrh_set_correlation lro_set_correlation
#define lrh_fft1_corr_begin lro_fft1_corr_begin
#define lrh_fft1_corr_finish lro_fft1_corr_finish
#define lrh_get_slowcorr_tot_avgnum lro_get_slowcorr_tot_avgnum
#define lrh_set_filtercorr lro_set_filtercorr
#define lrh_set_liminfo lro_set_liminfo
#define lrh_set_mix1_selfreq lro_set_mix1_selfreq
#define lrh_timf1_write_async lro_timf1_write
static inline int shim_no_wait(lro_ctx *c) { (void)c; return 0; }
static inline int shim_no_register(lro_ctx *c, void *p, size_t n) { (void)c; (void)p; (void)n; return 0; }
static inline int shim_no_unregister(lro_ctx *c, void *p) { (void)c; (void)p; return 0; }
#define lrh_timf1_write_wait shim_no_wait
#define lrh_sync shim_no_wait        /* the oracle's calls are synchronous */
static inline int shim_no_stage_wait(lro_ctx *c, int stage) { (void)c; (void)stage; return 0; }
#define lrh_stage_wait shim_no_stage_wait
static inline int shim_no_stage_wait_lag(lro_ctx *c, int stage, int lag) { (void)c; (void)stage; (void)lag; return 0; }
#define lrh_stage_wait_lag shim_no_stage_wait_lag
static inline int shim_export_begin(lro_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt, int *ticket) { *ticket = 0; return lro_export(c, ring, dst, off, cnt); }
static inline int shim_export_end(lro_ctx *c, int ticket) { (void)c; (void)ticket; return 0; }
#define lrh_export_begin shim_export_begin
#define lrh_export_end shim_export_end
#define lrh_host_register shim_no_register
#define lrh_host_unregister shim_no_unregister
#endif
