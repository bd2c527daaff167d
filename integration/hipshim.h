/*
 * hipshim.h -- Linrad-side glue for liblinrad_hip.so (include/linrad_hip.h): what integration/linrad_hip.patch makes Linrad call.
 * Dropped into the Linrad source tree next to fft1.c together with hipshim.c; new fft1 version 21 ("HIP MI355X", fft_cntrl[21].gpu
 * = GPU_HIP) selects it the way versions 18 / 19 select clFFT / cuFFT (fft1var.c:62-63, globdef.h:33-34).
 */
#ifndef HIPSHIM_H
#define HIPSHIM_H
#ifndef GPU_HIP
#define GPU_HIP 3
#endif
extern int fft1_use_gpu;

/* hip_open: 0, a negative LRH_E* of the library, or 100 + n when the running configuration is one version 21 does not serve
   (103 the MMX back transform / second fft, 104 the correlation receiver, 106 NET_RXOUT_TIMF2 in the int16 format, 107 several mix1 channels,
   108 two RF channels together with spur removal): the caller ends with lirerr(1463) */
int  hip_open(void);                 /* wideband_dsp start, where create_clFFT_plan / cufftPlanMany are called (wcw.c:535-575) */
void hip_close(void);                /* wideband_dsp exit, where destroy_clFFT_plan is called (wcw.c:1174-1183)              */
void hip_timf1_new(int timf1p_pa, int nbytes);   /* finish_rx_read: one new block sits at timf1_char[timf1p_pa] (rxin.c:1425-1431) */
int  hip_fft1_b(int timf1p_ref, float *out, int gpu_handle_number);   /* fft1_b case 21 (fft1.c:3519-3553)                      */
void hip_fft1_c(void);               /* stand-ins for the stage functions of the same names                                     */
void hip_make_timf2(void);
void hip_first_noise_blanker(void);   /* also installs / removes the linear blanker's tables when hg.clever_bln_mode changes */
int  hip_fft1_update_liminfo(void);   /* selective limiter on the device-resident power spectra (sellim.c:738); 0: not taken (two channels: Linrad's own code on the summed spectra) */
int  hip_fft2_update_liminfo(void);   /* second limiter on the fft2 power sums (sellim.c:159); 0: not taken (hg.sellim_par1 != 2) */
void hip_make_fft2(void);
void hip_fft2_mix1_fixed(void);
void hip_fft1_mix1_fixed(void);       /* second fft off (Linrad's default): mix1.c:995, call site wcw.c:1712                 */
void hip_fft2_mix1_afc(void);         /* AFC on (default for weak-signal CW): mix1.c:863 / 1044, call sites wcw.c:1737 / 1700 */
void hip_fft1_mix1_afc(void);
void hip_compute_timf2_powersum(void);   /* wcw.c:80 (S/N meter)                                                              */
void hip_net_fft1(int timf1p_ref, int fft1_pa);   /* NET_RXOUT_FFT1: the retired batch into the host ring before the memcpy of wcw.c:1024-1043 */
void hip_net_timf2(int timf2_pt, int mm);         /* NET_RXOUT_TIMF2 / _FFT2: the span the network thread is about to walk (rxin.c:944, 1026) */
void hip_net_fft2(int fft2_pt, int count);
int  hip_store_new_spur(int pnt);     /* spur acquisition (spursub.c:619, 1247) on the device-resident fft2 spectra; remove_spur / swap_spurs */
int  hip_spur_phase_lock(int nx);
void hip_remove_spur(int ia);
void hip_swap_spurs(int ia, int ib);
extern int hip_sparse_rings;         /* -1 (default): cfg.fft2_float_sparse when no host code reads whole fft2 spectra; 0: every bin always (hipshim.c) */
struct lrh_ctx *hip_context(void);
struct lrh_ctx *hip_context_of(int channel);   /* two RF channels: one context each */    /* the context behind the hooks (diagnostics, tests)                                    */
#endif
