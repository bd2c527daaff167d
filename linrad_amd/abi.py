"""ctypes mirror of include/linrad_hip.h (the C ABI of liblinrad_hip.so).

This is the reference-side binding stub a maintainer would write for any C ABI: the
structs, the prototypes, and a thin object wrapper whose methods are named after the
reference stage functions they replace (fft1_b, fft1_c, make_timf2, first_noise_blanker,
make_fft2, fft2_mix1_fixed).  It is parametrised on the symbol prefix so that
tests can drive the CPU oracle (prefix ``lro``) through the very same calls.
"""
import ctypes as C
import numpy as np

LRH_OK, LRH_EINVAL, LRH_ENOMEM, LRH_EDEVICE, LRH_ESTATE, LRH_ERANGE, LRH_EINTERNAL = 0, -1, -2, -3, -4, -5, -6

(RING_TIMF1, RING_FFT1_FLOAT, RING_FFT1_SUMSQ, RING_FFT1_SLOWSUM, RING_TIMF2_FLOAT, RING_TIMF2_PWR,
 RING_FFT2_FLOAT, RING_FFT2_POWER, RING_FFT2_POWERSUM, RING_WG_WATERF, RING_TIMF3_FLOAT,
 RING_TIMF2_BLOCKPOWER, RING_FFT3, RING_BASEB_RAW, RING_FFT2_XYPOWER, RING_FFT2_XYSUM, RING_FFT1_CORRSUM, RING_FFT1_SLOWCORR,
 RING_FFT1_SLOWCORR_TOT) = range(19)
_RING_DTYPE = {RING_TIMF1: np.int16, RING_WG_WATERF: np.int16, RING_FFT1_SLOWCORR_TOT: np.float64}


class LrhConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int), ("device", C.c_int), ("rx_rf_channels", C.c_int),
        ("fft1_n", C.c_int), ("fft1_sinpow", C.c_int), ("fft1_gain", C.c_int), ("fft1_direction", C.c_int),
        ("fft_avg1num", C.c_int), ("fft_avg2num", C.c_int), ("timf1_bytes", C.c_int), ("max_fft1n", C.c_int),
        ("fft1_sumsq_bufsize", C.c_int), ("wg_xpoints", C.c_int), ("slowsum_fresh_recalc", C.c_int),
        ("bckfft_att_n", C.c_int), ("timf2pow_size", C.c_int),
        ("stupid_bln_mode", C.c_int), ("stupid_bln_factor", C.c_float), ("blnfit_range", C.c_int),
        ("blanker_pulsewidth", C.c_int), ("timf2_noise_floor_avgnum", C.c_int),
        ("blanker_info_update_interval", C.c_int), ("blanker_min_points", C.c_int), ("timf2_noise_floor", C.c_int),
        ("fft2_n", C.c_int), ("fft2_sinpow", C.c_int), ("max_fft2n", C.c_int), ("waterfall_avgnum", C.c_int),
        ("wf_first_xpoint", C.c_int), ("wf_xpixels", C.c_int), ("wf_mode", C.c_int), ("wf_lines", C.c_int),
        ("mix1_bandwidth_reduction_n", C.c_int), ("timf3_size", C.c_int), ("fftx_points_per_hz", C.c_float),
        ("mix1_lowest_fq", C.c_float), ("mix1_highest_fq", C.c_float),
        ("max_batch", C.c_int), ("second_fft_enable", C.c_int), ("timf2_blockpower_block", C.c_int),
        ("timf2_blockpower_size", C.c_int), ("timf1_frame_channels", C.c_int), ("timf1_channel_index", C.c_int),
        ("fft3_n", C.c_int), ("fft3_sinpow", C.c_int), ("mix2_n", C.c_int), ("max_fft3n", C.c_int),
        ("baseband_size", C.c_int), ("timf1_dword_input", C.c_int), ("sample_shift", C.c_int), ("blanker_channels", C.c_int), ("timf1_real_input", C.c_int), ("fft1_float_sparse", C.c_int), ("fft2_float_sparse", C.c_int),
    ]


class LrhPtrs(C.Structure):
    _fields_ = [
        ("timf1p_px", C.c_int), ("fft1_pa", C.c_int), ("fft1_na", C.c_int), ("fft1_nm", C.c_int),
        ("fft1_nb", C.c_int), ("fft1_pb", C.c_int),
        ("fft1_sumsq_pa", C.c_int), ("fft1_sumsq_counter", C.c_int), ("fft1_liminfo_cnt", C.c_int),
        ("fft1_sumsq_recalc", C.c_int),
        ("fft1_px", C.c_int), ("fft1_nx", C.c_int), ("timf2_pa", C.c_int),
        ("fft1_lowlevel_points", C.c_int), ("fft1_lowlevel_fraction", C.c_float),
        ("timf2p_fit", C.c_int), ("timf2_pn2", C.c_int), ("timf2_cleared_points_unused", C.c_int),
        ("timf2_blanker_points", C.c_int), ("blanker_info_update_counter", C.c_int),
        ("timf2_px", C.c_int), ("fft2_na", C.c_int), ("fft2_pa", C.c_int), ("fft2_nb", C.c_int), ("fft2_nm", C.c_int),
        ("wg_waterf_sum_counter", C.c_int), ("wg_waterf_ptr", C.c_int), ("fft2_liminfo_cnt", C.c_int),
        ("fft2_nx", C.c_int), ("timf3_pa", C.c_int), ("timf2_pb", C.c_int), ("timf2_blockpower_pa", C.c_int),
        ("timf3_px", C.c_int), ("fft3_pa", C.c_int), ("fft3_px", C.c_int), ("baseb_pa", C.c_int),
        ("timf3_py", C.c_int), ("reserved", C.c_int * 5),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class LrhBlankerState(C.Structure):
    _fields_ = [
        ("timf2_noise_floor", C.c_int), ("stupid_bln_limit", C.c_uint),
        ("timf2_despiked_pwr", C.c_float * 2), ("timf2_despiked_pwrinc", C.c_float * 2),
        ("stupid_blanker_rate", C.c_float), ("timf2_cleared_points", C.c_int),
        ("last_call_cleared", C.c_int), ("slow_path_calls", C.c_int),
        ("clever_bln_limit", C.c_uint), ("clever_blanker_rate", C.c_float), ("timf2_fitted_pulses", C.c_int),
        ("last_call_fitted", C.c_int), ("last_call_rejected", C.c_int), ("clever_serial_calls", C.c_int),
    ]


class LrhBlnInfo(C.Structure):
    _fields_ = [("size", C.c_int), ("rest", C.c_float), ("avgmax", C.c_float)]


class LrhBlankerTables(C.Structure):
    """lrh_blanker_tables: the linear blanker's pulse-response tables as init_blanker (buf.c:1786-2057) leaves them"""
    _fields_ = [("clever_bln_mode", C.c_int), ("clever_bln_factor", C.c_float), ("clever_bln_limit", C.c_uint),
                ("refpul_size", C.c_int), ("largest_blnfit", C.c_int), ("liminfo_amplitude_factor", C.c_float),
                ("bln", LrhBlnInfo * 7), ("refpulse", C.POINTER(C.c_float)), ("phasefunc", C.POINTER(C.c_float)),
                ("pulindex", C.POINTER(C.c_int))]


class LrhMix1State(C.Structure):
    _fields_ = [
        ("mix1_selfreq", C.c_double), ("mix1_point", C.c_int), ("mix1_old_point", C.c_int),
        ("mix1_phase", C.c_float), ("mix1_phase_step", C.c_float), ("mix1_phase_rot", C.c_float),
        ("mix1_old_phase", C.c_float),
    ]


class LrhAfc(C.Structure):
    """lrh_afc: the reference's per-transform AFC frequency tables (caller-owned rings)."""
    _fields_ = [("mix1_fq_mid", C.POINTER(C.c_float)), ("mix1_fq_slope", C.POINTER(C.c_float)),
                ("mix1_fq_curv", C.POINTER(C.c_float)), ("mix1_fq_start", C.POINTER(C.c_float)),
                ("baseband_bw_hz", C.c_float)]


class AfcTables:
    """numpy-backed lrh_afc with the reference's initial values (buf.c:1255-1258)."""

    def __init__(self, n, baseband_bw_hz):
        self.mid = np.full(n, -1, np.float32)
        self.start = np.full(n, -1, np.float32)
        self.slope = np.zeros(n, np.float32)
        self.curv = np.zeros(n, np.float32)
        fp = C.POINTER(C.c_float)
        self.c = LrhAfc(self.mid.ctypes.data_as(fp), self.slope.ctypes.data_as(fp), self.curv.ctypes.data_as(fp),
                        self.start.ctypes.data_as(fp), float(baseband_bw_hz))


class LrhSellim(C.Structure):
    """lrh_sellim: parameters of the selective limiter (fft1_update_liminfo, sellim.c:738)"""
    _fields_ = [("struct_size", C.c_int), ("sellim_maxlevel", C.c_int), ("spek_avgnum", C.c_int), ("fft1_blocktime", C.c_float),
                ("blanker_ston_fft1", C.c_float), ("sellim_par2", C.c_int), ("sellim_par3", C.c_int), ("sellim_par4", C.c_int),
                ("sellim_par5", C.c_int), ("sellim_par6", C.c_int), ("sellim_par7", C.c_int), ("sellim_par8", C.c_int),
                ("liminfo_group_points", C.c_int), ("fft1_first_point", C.c_int), ("fft1_last_point", C.c_int),
                ("fft1_first_inband", C.c_int), ("fft1_last_inband", C.c_int), ("baseband_bw_fftxpts", C.c_int),
                ("ston_scale", C.c_int), ("exact_stats", C.c_int), ("blanker_ston_fft2", C.c_float), ("fft2_blocktime", C.c_float),
                ("sellim_par1", C.c_int), ("fft1_desired", C.POINTER(C.c_float))]


def default_sellim(cfg, **kw):
    """hires_graph.c:1175-1189 defaults, uncalibrated end points (fft1.c:4615-4618), 16 noise-floor groups"""
    n1 = 1 << cfg.fft1_n
    s = LrhSellim(C.sizeof(LrhSellim), 12000, cfg.fft_avg1num * cfg.fft_avg2num, 0.0008, 4.0, 0, 0, 0, 0, 0, 0, 0,
                  n1 // 16, 0, n1 - 1, 0, n1 - 1, 40, 0, 1, 30.0, 0.0008 * (1 << cfg.fft2_n) / n1, 2, None)
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


class LrhSpur(C.Structure):
    """lrh_spur: PLL state of one tracked spur (spur.c / seldef.h globals spur_location, spur_flag, spur_freq, spur_d0pha ...)"""
    _fields_ = [("spur_location", C.c_int), ("spur_flag", C.c_int), ("spur_freq", C.c_float), ("spur_d0pha", C.c_float),
                ("spur_d1pha", C.c_float), ("spur_d2pha", C.c_float), ("spur_ampl", C.c_float), ("spur_noise", C.c_float),
                ("spur_avgd2", C.c_float)]


class LrhSynth(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("noise_sigma", C.c_float), ("ncarriers", C.c_int),
        ("carrier_bin", C.c_float * 16), ("carrier_amp", C.c_float * 16), ("fft_size", C.c_int),
        ("pulse_period", C.c_int), ("pulse_len", C.c_int), ("pulse_amp", C.c_float), ("chan_phase", C.c_float),
    ]


def default_config(fft1_n, fft2_n, **kw):
    """Reference defaults (weak-signal CW row of uivar.c:371; sizing rules of buf.c) for given sizes."""
    N1, N2 = 1 << fft1_n, 1 << fft2_n
    c = LrhConfig()
    c.struct_size = C.sizeof(LrhConfig)
    c.device = 0
    c.rx_rf_channels = 1
    c.fft1_n, c.fft1_sinpow, c.fft1_gain, c.fft1_direction = fft1_n, 2, 27, 1
    c.fft_avg1num, c.fft_avg2num = 5, 4
    c.max_fft1n = 8
    c.fft1_sumsq_bufsize = 8 * N1
    c.wg_xpoints, c.slowsum_fresh_recalc = N1 - 1, 2
    c.bckfft_att_n = 6
    c.timf2pow_size = 8 * max(N1, N2)
    c.timf1_bytes = 16 * N1 * 4
    c.stupid_bln_mode, c.stupid_bln_factor = 1, 5.0
    c.blnfit_range, c.blanker_pulsewidth = 48, 0
    c.timf2_noise_floor_avgnum, c.blanker_info_update_interval = 32, 4
    c.blanker_min_points = N2 // 3
    c.timf2_noise_floor = 200
    c.fft2_n, c.fft2_sinpow, c.max_fft2n = fft2_n, 2, 4
    c.waterfall_avgnum, c.wf_first_xpoint, c.wf_xpixels, c.wf_mode, c.wf_lines = 2, 0, min(N2, 1024), 1, 8
    c.mix1_bandwidth_reduction_n = 6
    c.timf3_size = 32 * max(8, N2 >> 6)
    c.fftx_points_per_hz, c.mix1_lowest_fq, c.mix1_highest_fq = 1.0, 0.0, float(N2)
    c.max_batch = 64
    c.second_fft_enable = 1
    c.timf2_blockpower_block, c.timf2_blockpower_size = 4 * 64, 1024
    c.fft3_n, c.fft3_sinpow, c.mix2_n, c.max_fft3n, c.baseband_size = 0, 2, 0, 8, 4096
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class LrhError(RuntimeError):
    pass


class StageAPI:
    """Object wrapper over the C ABI; ``prefix`` is 'lrh' (HIP product) or 'lro' (CPU oracle, tests only)."""

    def __init__(self, lib, prefix, cfg):
        self.lib, self.prefix, self.cfg = lib, prefix, cfg
        self._f = lambda name: getattr(lib, f"{prefix}_{name}")
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        self._proto("open", [C.POINTER(LrhConfig), C.POINTER(vp)])
        self._proto("close", [vp], None)
        self._proto("ptrs_init", [vp, C.POINTER(LrhPtrs)], None)
        self._proto("get_derived", [vp, ip, ip, ip, ip, ip])
        self._proto("set_filtercorr", [vp, fp])
        self._proto("set_liminfo", [vp, fp])
        self._proto("set_foldcorr", [vp, fp])
        self._proto("set_ch2_phasing", [vp, C.c_float, C.c_float])
        self._proto("set_waterfall_yfac", [vp, fp])
        self._proto("get_table", [vp, C.c_char_p, fp, C.c_int])
        self._proto("timf1_write", [vp, vp, C.c_int, C.c_int])
        self._proto("timf1_write_packed18", [vp, vp, C.c_int, C.c_int])
        if prefix == "lrh":                  # streaming producer calls: HIP library only
            self._proto("timf1_write_async", [vp, vp, C.c_int, C.c_int])
            self._proto("timf1_write_wait", [vp])
            self._proto("host_register", [vp, vp, C.c_size_t])
            self._proto("host_unregister", [vp, vp])
        self._proto("fft1_b", [vp, C.c_int, C.c_int, C.c_int, C.c_int])
        for n in ("fft1_c", "make_timf2", "make_fft2", "fft2_mix1_fixed", "fft1_mix1_fixed", "make_fft3_all", "fft3_mix2"):
            self._proto(n, [vp, C.POINTER(LrhPtrs), C.c_int])
        for n in ("fft2_mix1_afc", "fft1_mix1_afc"):
            self._proto(n, [vp, C.POINTER(LrhPtrs), C.c_int, C.POINTER(LrhAfc)])
        self._proto("first_noise_blanker", [vp, C.POINTER(LrhPtrs)])
        self._proto("blanker_begin", [vp, C.POINTER(LrhPtrs), ip])
        self._proto("blanker_finish", [vp, C.POINTER(LrhPtrs)])
        self._proto("blanker_weak_span", [vp, C.POINTER(C.c_size_t)])
        self._proto("fft2_xy_begin", [vp, C.POINTER(LrhPtrs), C.c_int, C.POINTER(C.c_size_t)])
        self._proto("set_correlation", [vp, C.c_int])
        self._proto("fft1_corr_begin", [vp, C.POINTER(LrhPtrs), C.c_int, C.POINTER(C.c_size_t)])
        self._proto("fft1_corr_finish", [vp, C.POINTER(LrhPtrs), C.c_int])
        self._proto("get_slowcorr_tot_avgnum", [vp, ip])
        self._proto("fft2_xy_finish", [vp, C.POINTER(LrhPtrs), C.c_int])
        self._proto("set_pol", [vp, C.c_float, C.c_float, C.c_float])
        self._proto("set_combine_weights", [vp, C.c_float, C.c_float, C.c_float, C.c_float])
        self._proto("mix2_pol_begin", [vp, C.POINTER(LrhPtrs), C.c_int, C.POINTER(C.c_size_t)])
        self._proto("exchange_ptr", [vp, C.c_int, C.POINTER(vp)])
        self._proto("exchange_read", [vp, C.c_int, fp, C.c_size_t, C.c_size_t])
        self._proto("exchange_write", [vp, C.c_int, fp, C.c_size_t, C.c_size_t])
        self._proto("compute_timf2_powersum", [vp, C.POINTER(LrhPtrs)])
        self._proto("set_bg_filterfunc", [vp, fp])
        self._proto("set_basebraw_fir", [vp, fp, C.c_int])
        self._proto("fft1_update_liminfo", [vp, C.POINTER(LrhPtrs), C.POINTER(LrhSellim)])
        self._proto("set_blanker_tables", [vp, C.POINTER(LrhBlankerTables)])
        self._proto("spur_config", [vp, C.c_int, C.c_int, fp])
        self._proto("spur_set", [vp, C.c_int, C.POINTER(LrhSpur), fp, fp, ip])
        self._proto("spur_get", [vp, C.c_int, C.POINTER(LrhSpur), ip])
        self._proto("spur_acquire", [vp, C.POINTER(LrhPtrs), C.c_int, ip])
        self._proto("spur_search_config", [vp, C.c_int, C.c_int])
        self._proto("spur_search_get", [vp, fp, fp, ip, ip])
        self._proto("get_liminfo", [vp, fp])
        self._proto("fft2_update_liminfo", [vp, C.POINTER(LrhPtrs), C.POINTER(LrhSellim)])
        self._proto("wideband_limiter", [vp, C.POINTER(LrhSellim), C.c_int])
        self._proto("get_liminfo_amplitude_factor", [vp, fp])
        self._proto("set_liminfo_amplitude_factor", [vp, C.c_float])
        self._proto("set_mix1_selfreq", [vp, C.c_double])
        self._proto("get_mix1_state", [vp, C.POINTER(LrhMix1State)])
        self._proto("wideband_dsp", [vp, C.POINTER(LrhPtrs), C.c_int, C.c_int])
        self._proto("export", [vp, C.c_int, vp, C.c_size_t, C.c_size_t])
        self._proto("get_blanker_state", [vp, C.POINTER(LrhBlankerState)])
        self._proto("export_timf2_net", [vp, fp, C.c_int, C.c_int, C.c_float, C.c_float])
        if prefix == "lrh":
            self._proto("export_fft1_net", [vp, fp, C.c_int, C.c_int])
        self.ctx = vp()
        rc = self._f("open")(C.byref(cfg), C.byref(self.ctx))
        if rc != 0:
            raise LrhError(f"{prefix}_open failed rc={rc}")
        self.p = LrhPtrs()
        self._f("ptrs_init")(self.ctx, C.byref(self.p))
        d = [C.c_int() for _ in range(5)]
        self._f("get_derived")(self.ctx, *[C.byref(x) for x in d])
        (self.fft1_interleave_points, self.fft2_interleave_points, self.mix1_size,
         self.mix1_interleave_points, self.timf3_block) = [x.value for x in d]
        self.N1, self.N2 = 1 << cfg.fft1_n, 1 << cfg.fft2_n
        # fft3 / mix2 sizes (baseb_graph.c:636-645)
        self.fft3_interleave_points = 0
        if cfg.fft3_n:
            ratio = 0.0 if cfg.fft3_sinpow == 0 else (0.625 if cfg.fft3_sinpow == 9 else 0.8 if cfg.fft3_sinpow == 8 else
                                                     2 * np.arcsin(0.5 ** (1.0 / cfg.fft3_sinpow)) / np.pi)
            m2 = 1 << cfg.mix2_n
            mi = int(np.float32(ratio) * m2) & ~1
            self.mix2_interleave_points = mi
            self.fft3_interleave_points = mi * ((1 << cfg.fft3_n) // m2)
        self.timf1_blockbytes = (self.N1 - self.fft1_interleave_points) * (8 if cfg.timf1_dword_input else 4) * max(1, cfg.timf1_frame_channels)

    def _proto(self, name, argtypes, restype=C.c_int):
        f = self._f(name)
        f.argtypes, f.restype = argtypes, restype

    def _chk(self, rc, what):
        if rc != 0:
            raise LrhError(f"{self.prefix}_{what} rc={rc}")

    def close(self):
        if self.ctx:
            self._f("close")(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _fptr(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    # ---- tables
    def set_filtercorr(self, fc=None):
        if fc is None:
            self._chk(self._f("set_filtercorr")(self.ctx, None), "set_filtercorr")
        else:
            fc = np.ascontiguousarray(fc, np.float32)
            assert fc.size == 2 * self.N1
            self._chk(self._f("set_filtercorr")(self.ctx, self._fptr(fc)), "set_filtercorr")

    def set_foldcorr(self, foldcorr):
        """fft1_foldcorr (N1 complex as 2*N1 floats) or None: I/Q mirror-image calibration (fft1.c:3598-3658)."""
        if foldcorr is None:
            self._chk(self._f("set_foldcorr")(self.ctx, None), "set_foldcorr")
        else:
            t = np.ascontiguousarray(foldcorr, np.float32)
            assert t.size == 2 * self.N1
            self._chk(self._f("set_foldcorr")(self.ctx, self._fptr(t)), "set_foldcorr")

    # ---- two coupled RF channels (cfg.blanker_channels = 2): see include/linrad_hip.h
    X_PWR, X_STAT, X_BINS, X_POL, X_WEAK, X_SPEC = 0, 1, 2, 3, 4, 5

    def ptrs_copy(self):
        """the pointer state as it is now (`at` of lrh_fft2_xy_begin / finish: taken before make_fft2)"""
        q = LrhPtrs()
        C.memmove(C.byref(q), C.byref(self.p), C.sizeof(LrhPtrs))
        return q

    def set_correlation(self, on=True):
        """correlation spectrum of two coupled channels (genparm[FFT1_CORRELATION_SPECTRUM] = 1): fft1_corrsum / fft1_slowcorr / _tot"""
        self._chk(self._f("set_correlation")(self.ctx, int(bool(on))), "set_correlation")

    def fft1_corr_begin(self, at, batch=1):
        n = C.c_size_t()
        self._chk(self._f("fft1_corr_begin")(self.ctx, C.byref(at), batch, C.byref(n)), "fft1_corr_begin")
        return n.value

    def fft1_corr_finish(self, at, batch=1):
        self._chk(self._f("fft1_corr_finish")(self.ctx, C.byref(at), batch), "fft1_corr_finish")

    def slowcorr_tot_avgnum(self):
        n = C.c_int()
        self._chk(self._f("get_slowcorr_tot_avgnum")(self.ctx, C.byref(n)), "get_slowcorr_tot_avgnum")
        return n.value

    def fft2_xy_begin(self, at, batch=1):
        n = C.c_size_t()
        self._chk(self._f("fft2_xy_begin")(self.ctx, C.byref(at), batch, C.byref(n)), "fft2_xy_begin")
        return n.value

    def set_pol(self, c1, c2, c3):
        """pg.c1..c3: the polarisation fft3_mix2 turns the channel pair into (mix2.c:340-343)"""
        self._chk(self._f("set_pol")(self.ctx, float(c1), float(c2), float(c3)), "set_pol")

    def set_combine_weights(self, wa, wb=0j):
        """this channel's complex weights in the two coherent sums A = sum_c wa_c X_c, B = sum_c wb_c X_c (phased array)"""
        wa, wb = complex(wa), complex(wb)
        self._chk(self._f("set_combine_weights")(self.ctx, wa.real, wa.imag, wb.real, wb.imag), "set_combine_weights")

    def mix2_pol_begin(self, batch=1):
        n = C.c_size_t()
        self._chk(self._f("mix2_pol_begin")(self.ctx, C.byref(self.p), batch, C.byref(n)), "mix2_pol_begin")
        return n.value

    def fft2_xy_finish(self, at, batch=1):
        self._chk(self._f("fft2_xy_finish")(self.ctx, C.byref(at), batch), "fft2_xy_finish")

    def blanker_begin(self):
        n = C.c_int()
        self._chk(self._f("blanker_begin")(self.ctx, C.byref(self.p), C.byref(n)), "blanker_begin")
        return n.value

    def blanker_weak_span(self):
        """floats per slot of X_WEAK the coming first_noise_blanker wants gathered (linear blanker on two coupled channels; else 0)"""
        n = C.c_size_t()
        self._chk(self._f("blanker_weak_span")(self.ctx, C.byref(n)), "blanker_weak_span")
        return n.value

    def blanker_finish(self):
        self._chk(self._f("blanker_finish")(self.ctx, C.byref(self.p)), "blanker_finish")

    def exchange_read(self, which, count, off=0):
        out = np.empty(count, np.float32)
        self._chk(self._f("exchange_read")(self.ctx, which, self._fptr(out), off, count), "exchange_read")
        return out

    def exchange_write(self, which, data, off=0):
        data = np.ascontiguousarray(data, np.float32)
        self._chk(self._f("exchange_write")(self.ctx, which, self._fptr(data), off, data.size), "exchange_write")

    def exchange_ptr(self, which):
        ptr = C.c_void_p()
        self._chk(self._f("exchange_ptr")(self.ctx, which, C.byref(ptr)), "exchange_ptr")
        return ptr.value

    def set_ch2_phasing(self, c1, c2):
        """pg_ch2_c1 / pg_ch2_c2 for the context that carries the second RF channel (fft1.c:4064-4080)."""
        self._chk(self._f("set_ch2_phasing")(self.ctx, float(c1), float(c2)), "set_ch2_phasing")

    def set_liminfo(self, lim):
        lim = np.ascontiguousarray(lim, np.float32)
        assert lim.size == self.N1
        self._chk(self._f("set_liminfo")(self.ctx, self._fptr(lim)), "set_liminfo")

    def fft1_update_liminfo(self, par):
        """one run of the selective limiter on the device-resident spectra (sellim.c:738-1157), see include/linrad_hip.h"""
        self._chk(self._f("fft1_update_liminfo")(self.ctx, C.byref(self.p), C.byref(par)), "fft1_update_liminfo")

    def wideband_limiter(self, par=None, fft2_too=False):
        """the limiter calls of wideband_dsp's loop (wcw.c:1124-1133) inside every round of wideband_dsp; None: off"""
        self._chk(self._f("wideband_limiter")(self.ctx, C.byref(par) if par is not None else None, int(bool(fft2_too))), "wideband_limiter")

    def fft2_update_liminfo(self, par):
        """fft2_update_liminfo (sellim.c:159, par1 = 2) on the device-resident fft2 power sums"""
        self._chk(self._f("fft2_update_liminfo")(self.ctx, C.byref(self.p), C.byref(par)), "fft2_update_liminfo")

    def liminfo_amplitude_factor(self):
        f = C.c_float()
        self._chk(self._f("get_liminfo_amplitude_factor")(self.ctx, C.byref(f)), "get_liminfo_amplitude_factor")
        return f.value

    def set_liminfo_amplitude_factor(self, f):
        self._chk(self._f("set_liminfo_amplitude_factor")(self.ctx, float(f)), "set_liminfo_amplitude_factor")

    def get_liminfo(self):
        out = np.empty(self.N1, np.float32)
        self._chk(self._f("get_liminfo")(self.ctx, self._fptr(out)), "get_liminfo")
        return out

    def set_blanker_tables(self, bln=None, refpulse=None, phasefunc=None, pulindex=None, largest_blnfit=0, clever_bln_factor=10.0,
                           clever_bln_limit=0, clever_bln_mode=1, liminfo_amplitude_factor=1.0):
        """install the linear blanker's tables (bln: rows of (size, rest, avgmax)); no arguments: clever blanker off"""
        if bln is None:
            self._chk(self._f("set_blanker_tables")(self.ctx, None), "set_blanker_tables")
            return
        t = LrhBlankerTables()
        rp, pf = np.ascontiguousarray(refpulse, np.float32), np.ascontiguousarray(phasefunc, np.float32)
        pi = np.ascontiguousarray(pulindex, np.int32)
        t.refpul_size = pf.size // 2
        assert rp.size == 2 * 256 * t.refpul_size and pi.size == 256
        t.clever_bln_mode, t.clever_bln_factor, t.clever_bln_limit = int(clever_bln_mode), float(clever_bln_factor), int(clever_bln_limit)
        t.largest_blnfit, t.liminfo_amplitude_factor = int(largest_blnfit), float(liminfo_amplitude_factor)
        for i, row in enumerate(bln):
            t.bln[i].size, t.bln[i].rest, t.bln[i].avgmax = int(row[0]), float(row[1]), float(row[2])
        t.refpulse, t.phasefunc, t.pulindex = self._fptr(rp), self._fptr(pf), pi.ctypes.data_as(C.POINTER(C.c_int))
        self._chk(self._f("set_blanker_tables")(self.ctx, C.byref(t)), "set_blanker_tables")

    def spur_config(self, max_spurs, spur_speknum, spur_spectra):
        t = np.ascontiguousarray(spur_spectra, np.float32)
        assert t.size == 2048
        self._chk(self._f("spur_config")(self.ctx, int(max_spurs), int(spur_speknum), self._fptr(t)), "spur_config")

    def spur_set(self, spurs, table, signal, ind):
        """hand over the control plane's spurs: list of LrhSpur + per-spur histories (see include/linrad_hip.h)"""
        arr = (LrhSpur * max(1, len(spurs)))(*spurs)
        table, signal = np.ascontiguousarray(table, np.float32), np.ascontiguousarray(signal, np.float32)
        ind = np.ascontiguousarray(ind, np.int32)
        self._chk(self._f("spur_set")(self.ctx, len(spurs), arr, self._fptr(table), self._fptr(signal), ind.ctypes.data_as(C.POINTER(C.c_int))), "spur_set")

    def spur_acquire(self, pnt):
        """store_new_spur + spur_phase_lock on the resident fft2 spectra for the seven bins from `pnt` (spursub.c:619, 1247); True: locked and now tracked"""
        locked = C.c_int()
        self._chk(self._f("spur_acquire")(self.ctx, C.byref(self.p), int(pnt), C.byref(locked)), "spur_acquire")
        return bool(locked.value)

    def spur_search_config(self, first_point, last_point):
        """the search for new spurs on the resident power rows: make_fft2 keeps the sums over 3 spur_speknum transforms and cleans the
        finished search spectrum (fft2.c:673-699, spursearch_spectrum_cleanup spursub.c:40); (0, 0): off"""
        self._first_last = (int(first_point), int(last_point))
        self._chk(self._f("spur_search_config")(self.ctx, int(first_point), int(last_point)), "spur_search_config")

    def spur_search_get(self, spectrum=True):
        """(spursearch_spectrum[first .. last] or None, spur_search_threshold, search spectra completed so far, spursearch_sum_counter)"""
        a, b = self._first_last
        out = np.zeros(b - a + 1, np.float32) if spectrum else None
        thr, done, cnt = C.c_float(), C.c_int(), C.c_int()
        self._chk(self._f("spur_search_get")(self.ctx, self._fptr(out) if spectrum else None, C.byref(thr), C.byref(done), C.byref(cnt)), "spur_search_get")
        return out, thr.value, done.value, cnt.value

    def spur_get(self, max_spurs=16):
        arr = (LrhSpur * max_spurs)()
        n = C.c_int()
        self._chk(self._f("spur_get")(self.ctx, max_spurs, arr, C.byref(n)), "spur_get")
        return [arr[i] for i in range(n.value)]

    def set_waterfall_yfac(self, y=None):
        if y is None:
            self._chk(self._f("set_waterfall_yfac")(self.ctx, None), "set_waterfall_yfac")
        else:
            y = np.ascontiguousarray(y, np.float32)
            assert y.size == self.N1
            self._chk(self._f("set_waterfall_yfac")(self.ctx, self._fptr(y)), "set_waterfall_yfac")

    def get_table(self, name, count):
        out = np.zeros(count, np.float32)
        n = self._f("get_table")(self.ctx, name.encode(), self._fptr(out), count)
        if n < 0:
            raise LrhError(f"get_table({name}) rc={n}")
        return out[:n]

    # ---- producer
    def timf1_write(self, iq, byte_offset=0):
        iq = np.ascontiguousarray(iq, np.int32 if self.cfg.timf1_dword_input else np.int16)
        self._chk(self._f("timf1_write")(self.ctx, iq.ctypes.data_as(C.c_void_p), int(byte_offset), int(iq.nbytes)),
                  "timf1_write")

    def timf1_write_async(self, iq, byte_offset=0):
        """producer copy without the host wait (lrh_timf1_write_async); `iq` must stay alive and untouched until timf1_write_wait()"""
        assert iq.flags["C_CONTIGUOUS"]
        self._chk(self._f("timf1_write_async")(self.ctx, iq.ctypes.data_as(C.c_void_p), int(byte_offset), int(iq.nbytes)), "timf1_write_async")

    def timf1_write_wait(self):
        self._chk(self._f("timf1_write_wait")(self.ctx), "timf1_write_wait")

    def host_register(self, arr):
        self._chk(self._f("host_register")(self.ctx, arr.ctypes.data_as(C.c_void_p), int(arr.nbytes)), "host_register")

    def host_unregister(self, arr):
        self._chk(self._f("host_unregister")(self.ctx, arr.ctypes.data_as(C.c_void_p)), "host_unregister")

    def timf1_write_packed18(self, packed, byte_offset=0):
        """One read of an 18-bit .raw recording: packed bytes -> int32 ring (expand_rawdat, csplit.c:20-73)."""
        packed = np.ascontiguousarray(packed, np.uint8)
        self._chk(self._f("timf1_write_packed18")(self.ctx, packed.ctypes.data_as(C.c_void_p), int(byte_offset),
                                                  int(packed.nbytes)), "timf1_write_packed18")

    # ---- stages (names = reference functions)
    def fft1_b(self, batch=1, handle=0):
        """fft1_b for `batch` blocks (handle = gpu_handle_number: 0 own thread, 1..6 worker), then the caller-side pointer
        advance of wcw.c:1036-1047."""
        p = self.p
        self._chk(self._f("fft1_b")(self.ctx, handle, p.timf1p_px, p.fft1_pa, batch), "fft1_b")
        block = 2 * self.N1
        p.timf1p_px = (p.timf1p_px + batch * self.timf1_blockbytes) & (self.cfg.timf1_bytes - 1)
        p.fft1_pa = (p.fft1_pa + batch * block) & (self.cfg.max_fft1n * block - 1)
        p.fft1_na = p.fft1_pa // block
        p.fft1_nm = min(p.fft1_nm + batch, self.cfg.max_fft1n - 1)

    def fft1_c(self, batch=1):
        self._chk(self._f("fft1_c")(self.ctx, C.byref(self.p), batch), "fft1_c")

    def make_timf2(self, batch=1):
        self._chk(self._f("make_timf2")(self.ctx, C.byref(self.p), batch), "make_timf2")

    def first_noise_blanker(self):
        self._chk(self._f("first_noise_blanker")(self.ctx, C.byref(self.p)), "first_noise_blanker")

    def make_fft2(self, batch=1):
        self._chk(self._f("make_fft2")(self.ctx, C.byref(self.p), batch), "make_fft2")

    def fft2_mix1_fixed(self, batch=1):
        self._chk(self._f("fft2_mix1_fixed")(self.ctx, C.byref(self.p), batch), "fft2_mix1_fixed")

    def fft1_mix1_fixed(self, batch=1):
        self._chk(self._f("fft1_mix1_fixed")(self.ctx, C.byref(self.p), batch), "fft1_mix1_fixed")

    def fft2_mix1_afc(self, afc, batch=1):
        self._chk(self._f("fft2_mix1_afc")(self.ctx, C.byref(self.p), batch, C.byref(afc.c)), "fft2_mix1_afc")

    def fft1_mix1_afc(self, afc, batch=1):
        self._chk(self._f("fft1_mix1_afc")(self.ctx, C.byref(self.p), batch, C.byref(afc.c)), "fft1_mix1_afc")

    def make_fft3_all(self, batch=1):
        self._chk(self._f("make_fft3_all")(self.ctx, C.byref(self.p), batch), "make_fft3_all")

    def fft3_mix2(self, batch=1):
        self._chk(self._f("fft3_mix2")(self.ctx, C.byref(self.p), batch), "fft3_mix2")

    def set_bg_filterfunc(self, f):
        f = np.ascontiguousarray(f, np.float32)
        assert f.size == (1 << self.cfg.fft3_n)
        self._chk(self._f("set_bg_filterfunc")(self.ctx, self._fptr(f)), "set_bg_filterfunc")

    def set_basebraw_fir(self, fir=None):
        """bg.mixer_mode = 2: fft3_mix2 decimates timf3 with this FIR (mix2.c:217-246); None: back to the filter on fft3's bins"""
        if fir is None:
            self._chk(self._f("set_basebraw_fir")(self.ctx, None, 0), "set_basebraw_fir")
        else:
            fir = np.ascontiguousarray(fir, np.float32)
            self._chk(self._f("set_basebraw_fir")(self.ctx, self._fptr(fir), int(fir.size)), "set_basebraw_fir")

    def fft3_available(self):
        """transforms make_fft3_all may run now (do_fft3 loop condition, fft3.c:54-55)"""
        c, p = self.cfg, self.p
        n3 = 1 << c.fft3_n
        have = (p.timf3_pa - p.timf3_px + c.timf3_size) & (c.timf3_size - 1)
        if have < 2 * n3:
            return 0
        new3 = n3 - self.fft3_interleave_points
        return 1 + (have - 2 * n3) // (2 * new3)

    def compute_timf2_powersum(self):
        self._chk(self._f("compute_timf2_powersum")(self.ctx, C.byref(self.p)), "compute_timf2_powersum")

    def fft2_available(self):
        """number of fft2 transforms the released timf2 data allows (wcw.c:265-275)"""
        p = self.p
        size = 4 * self.cfg.timf2pow_size
        avail = (p.timf2_pn2 - p.timf2_px + size) & (size - 1)
        if avail < 4 * self.N2:
            return 0
        return 1 + (avail - 4 * self.N2) // (4 * (self.N2 - self.fft2_interleave_points))

    def set_mix1_selfreq(self, fq):
        self._chk(self._f("set_mix1_selfreq")(self.ctx, float(fq)), "set_mix1_selfreq")

    def mix1_state(self):
        st = LrhMix1State()
        self._chk(self._f("get_mix1_state")(self.ctx, C.byref(st)), "get_mix1_state")
        return st

    def set_exchange(self, fn):
        """register the cross-channel exchange function of two coupled channels (lrh_set_exchange): fn(which, op, ptr, count, stream, own) -> 0
        (own: where the caller's contribution to a gather lies when it is not in its slot, else None);
        None removes it.  The ctypes thunk is kept alive on the receiver."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
        f = self._f("set_exchange")
        f.argtypes, f.restype = [C.c_void_p, proto, C.c_void_p], C.c_int
        if fn is None:
            self._xthunk = proto()
        else:
            def thunk(user, which, op, ptr, count, stream, own):
                try:
                    return int(fn(which, op, ptr, count, stream, own) or 0)
                except Exception:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            self._xthunk = proto(thunk)
        self._chk(f(self.ctx, self._xthunk, None), "set_exchange")

    def wideband_dsp(self, nblocks, batch):
        self._chk(self._f("wideband_dsp")(self.ctx, C.byref(self.p), nblocks, batch), "wideband_dsp")

    # ---- outputs
    def ring_size(self, ring):
        c = self.cfg
        return {RING_TIMF1: c.timf1_bytes // 2, RING_FFT1_FLOAT: c.max_fft1n * 2 * self.N1,
                RING_FFT1_SUMSQ: c.fft1_sumsq_bufsize, RING_FFT1_SLOWSUM: self.N1,
                RING_TIMF2_FLOAT: 4 * c.timf2pow_size, RING_TIMF2_PWR: c.timf2pow_size,
                RING_FFT2_FLOAT: c.max_fft2n * 2 * self.N2, RING_FFT2_POWER: c.max_fft2n * self.N2,
                RING_FFT2_POWERSUM: self.N2, RING_WG_WATERF: c.wf_lines * c.wf_xpixels,
                RING_TIMF3_FLOAT: c.timf3_size, RING_TIMF2_BLOCKPOWER: c.timf2_blockpower_size,
                RING_FFT3: c.max_fft3n * 2 * (1 << c.fft3_n) if c.fft3_n else 0,
                RING_BASEB_RAW: 2 * c.baseband_size, RING_FFT2_XYPOWER: c.max_fft2n * 4 * self.N2,
                RING_FFT2_XYSUM: 4 * self.N2, RING_FFT1_CORRSUM: 2 * c.fft1_sumsq_bufsize, RING_FFT1_SLOWCORR: 2 * self.N1,
                RING_FFT1_SLOWCORR_TOT: 2 * self.N1}[ring]

    def export(self, ring, offset=0, count=None):
        if count is None:
            count = self.ring_size(ring) - offset
        out = np.zeros(count, _RING_DTYPE.get(ring, np.float32))
        self._chk(self._f("export")(self.ctx, ring, out.ctypes.data_as(C.c_void_p), offset, count), "export")
        return out

    def export_timf2_net(self, timf2_pt, count, map65_gain=1.0, map65_strong=1.0):
        """NET_RXOUT_TIMF2 payload (rxin.c:944-966): gain * (weak + strong_scale * strong) as `count` complex floats"""
        out = np.empty(2 * count, np.float32)
        self._chk(self._f("export_timf2_net")(self.ctx, self._fptr(out), int(timf2_pt), int(count), float(map65_gain), float(map65_strong)), "export_timf2_net")
        return out

    def export_fft1_net(self, timf1p_ref, batch=1):
        """NET_RXOUT_FFT1 payload (wcw.c:1024-1043): `batch` transforms as fft1_b leaves them, before fft1_c's filter correction"""
        out = np.empty(2 * self.N1 * batch, np.float32)
        self._chk(self._f("export_fft1_net")(self.ctx, self._fptr(out), int(timf1p_ref), int(batch)), "export_fft1_net")
        return out

    def blanker_state(self):
        st = LrhBlankerState()
        self._chk(self._f("get_blanker_state")(self.ctx, C.byref(st)), "get_blanker_state")
        return st
