"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- one RF channel per rank, all-reduce of the channel power sums."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from linrad_amd import abi
    from linrad_amd.lib import synth_defaults, synth_iq
    from linrad_amd.multichan import channel_of_rank, cross_channel_power_sum, newest_sumsq_block
    from linrad_amd.workload import chain_config, strong_liminfo
    from oracle_binding import open_oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = chain_config(fft1_n=10, fft2_n=8, batch=8)
    s = synth_defaults(1 << cfg.fft1_n, channel_of_rank(rank))
    rx = open_oracle(cfg)                       # CPU stand-in for the per-GPU receiver; same StageAPI
    rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
    rx.set_liminfo(strong_liminfo(s, cfg.fft1_n))
    rx.wideband_dsp(10, 5)                      # completes two averaging periods (fft_avg1num = 5)
    own = rx.export(abi.RING_FFT1_SUMSQ, newest_sumsq_block(rx), rx.N1)
    t = torch.from_numpy(own.copy())
    cross_channel_power_sum(t, dist)
    gathered = [torch.zeros(rx.N1) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(own.copy()))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, t.numpy(), [g.numpy() for g in gathered]))


def test_two_channel_power_sum_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, sum0, parts0), (_, sum1, parts1) = res
    # both ranks hold the same cross-channel sum, equal to the two per-channel spectra added (fft1.c:4145)
    assert np.array_equal(sum0, sum1)
    assert np.allclose(sum0, parts0[0] + parts0[1], rtol=1e-6)
    # the channels really are different signals (independent noise, rotated carriers)
    assert not np.allclose(parts0[0], parts0[1], rtol=1e-3)
    assert np.all(sum0 >= 0) and np.count_nonzero(sum0) >= sum0.size - 2      # the two edge bins are filtered to zero


def _coupled_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from linrad_amd import abi
    from linrad_amd.multichan import run_coupled
    from oracle_binding import open_oracle
    from refcases import lrh_config, twochan_case

    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, frames, lim = twochan_case("twochan_n10")
    iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 2 * rank:2 * rank + 2]).ravel()
    cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=rank)
    rx = open_oracle(cfg)
    rx.timf1_write(iq)
    rx.set_liminfo(lim)
    if rank == 1:
        rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
    run_coupled(rx, d["nblk"], 1, dist, mix1=False)            # the golden's call pattern: block by block
    bs = rx.blanker_state()
    out = dict(timf2=rx.export(abi.RING_TIMF2_FLOAT), nf=bs.timf2_noise_floor, lim=bs.stupid_bln_limit, fit=rx.p.timf2p_fit,
               fft2_na=rx.p.fft2_na, fft2=rx.export(abi.RING_FFT2_FLOAT))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_coupled_two_channel_blanker_over_gloo():
    """One channel per rank, the blanker coupled through two all-reduces per call (multichan.coupled_blanker): every rank
    must end with the compiled two-channel reference's blanker state and its channel of the blanked timf2 ring."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coupled_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "twochan_n10.npz"))
    it = g["bln_itrace"].reshape(-1, 16)
    fit = int(it[-1, 1])
    gt = g["bln_timf2_float"].reshape(-1, 2, 2, 2)
    assert res[0]["nf"] == res[1]["nf"] and res[0]["lim"] == res[1]["lim"] and res[0]["fit"] == res[1]["fit"] == fit
    assert abs(res[0]["nf"] - int(it[-1, 12])) <= 1
    for ch in (0, 1):
        t = res[ch]["timf2"].reshape(-1, 2, 2)[:fit]
        ref = gt[:fit, :, ch, :]
        same_cleared = np.array_equal((t[:, 0, :] == 0).all(axis=1), (ref[:, 0, :] == 0).all(axis=1))
        if res[0]["nf"] == int(it[-1, 12]):
            assert same_cleared
            assert np.linalg.norm(t - ref) <= 2e-6 * np.linalg.norm(ref)
        assert np.count_nonzero(res[ch]["fft2"]) > 0            # the chain went on behind the blanker (make_fft2 on the released data)


def _clever2_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import cleverlib
    from linrad_amd import abi
    from linrad_amd.multichan import coupled_blanker
    from oracle_binding import open_oracle
    from refcases import clever2_case, lrh_config

    dist.init_process_group("gloo", rank=rank, world_size=world)
    name = "clever2_n10"
    g = cleverlib.load(name)
    d, cl, frames, lim, des = clever2_case(name)
    bi = g["bln_ints"]
    d = dict(d, pulsewidth=int(bi[1]), blnfit_range=int(bi[3]))
    iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 2 * rank:2 * rank + 2]).ravel()
    rx = open_oracle(lrh_config(d, iq, blanker_channels=2, timf1_channel_index=rank))
    rx.timf1_write(iq)
    rx.set_liminfo(lim)
    cleverlib.install_tables(rx, g, d["noise_floor"])
    fitted = []
    for _ in range(d["nblk"]):
        rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
        coupled_blanker(rx, dist)
        fitted.append(rx.blanker_state().timf2_fitted_pulses)
    out = dict(timf2=rx.export(abi.RING_TIMF2_FLOAT), fitted=fitted, fit=rx.p.timf2p_fit, pa=rx.p.timf2_pa, n1=rx.N1)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_coupled_linear_blanker_over_gloo():
    """The linear blanker on two channels, one per rank: power-sum all-reduce plus the all-gather of the channels' weak samples
    (multichan.coupled_blanker); both ranks run the same fits and end with the compiled two-channel reference's pulse counts and
    their channel of its timf2 ring."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clever2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "clever2_n10.npz"))
    it, tr = g["itrace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    assert res[0]["fitted"] == res[1]["fitted"] == tr[:, 10].astype(int).tolist() and max(res[0]["fitted"]) > 5
    assert res[0]["fit"] == res[1]["fit"] == int(it[-1, 1])
    gt = g["timf2_float"].reshape(-1, 2, 2, 2)
    keep = np.ones(gt.shape[0], bool)
    keep[(res[0]["pa"] // 4 + np.arange(res[0]["n1"] // 2)) % keep.size] = False
    for ch in (0, 1):
        t = res[ch]["timf2"].reshape(-1, 2, 2)[keep]
        ref = gt[keep][:, :, ch, :]
        assert np.linalg.norm(t - ref) <= 2e-6 * np.linalg.norm(ref)


def _chain_worker(rank, world, port, q, entry="stages"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from linrad_amd import abi
    from linrad_amd.multichan import run_coupled
    from oracle_binding import open_oracle
    from refcases import lrh_config, twochan_case

    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, frames, lim = twochan_case("twochan_n10", chain=True)
    iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 2 * rank:2 * rank + 2]).ravel()
    cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=rank)
    rx = open_oracle(cfg)
    rx.timf1_write(iq)
    rx.set_liminfo(lim)
    rx.set_mix1_selfreq(d["fq"])
    g = np.load(os.path.join(ROOT, "tests", "golden", "twochan_n10_chain.npz"))
    rx.set_bg_filterfunc(g["bg_filterfunc"])
    rx.set_pol(*d["pol"])
    if rank == 1:
        rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
    if entry == "stages":
        run_coupled(rx, d["nblk"], 1, dist, xy=True, pol=True)
    else:                                                   # the same through one lrh_wideband_dsp call: the library asks for the collectives
        from linrad_amd.multichan import install_exchange
        install_exchange(rx, dist)
        rx.wideband_dsp(d["nblk"], 1)
    out = dict(baseb=rx.export(abi.RING_BASEB_RAW), baseb_pa=rx.p.baseb_pa, xyp=rx.export(abi.RING_FFT2_XYPOWER), xys=rx.export(abi.RING_FFT2_XYSUM), wf=rx.export(abi.RING_WG_WATERF),
               timf3=rx.export(abi.RING_TIMF3_FLOAT), fft2_na=rx.p.fft2_na, wptr=rx.p.wg_waterf_ptr)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


@pytest.mark.parametrize("entry", ["stages", "dsp"])
def test_two_channel_chain_over_gloo(entry):
    """The whole coupled chain with the collectives of linrad_amd/multichan.py: two all-reduces per blanker call and one
    all-gather of the new fft2 bins per make_fft2, one all-reduce of the polarisation sums per fft3_mix2.  Both ranks must end
    with the compiled two-channel reference's fft2_xypower / fft2_xysum and waterfall lines, each with its own channel of
    timf3; rank 0 with the reference's baseb_raw (wanted polarisation), rank 1 with baseb_raw_orthog."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, q, entry)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "twochan_n10_chain.npz"))
    fin = g["final"]
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))
    for r in (0, 1):
        assert res[r]["fft2_na"] == fin[7]
        assert rel(res[r]["xyp"], g["fft2_xypower"]) < 4e-6 and rel(res[r]["xys"], g["fft2_xysum"]) < 4e-6
    assert np.array_equal(res[0]["xyp"], res[1]["xyp"]) and np.array_equal(res[0]["wf"], res[1]["wf"])
    npix = g["wf_lines"].size // fin[11]
    lines = g["wf_lines"].reshape(-1, npix).astype(np.int32)
    wf = res[0]["wf"].reshape(-1, npix).astype(np.int32)
    # lines go downwards from the top of the ring (update_wg_waterf): line l sits at row (0 - l) mod wf_lines; the newest 8 survive
    for l in range(max(0, len(lines) - wf.shape[0]), len(lines)):
        diff = np.abs(wf[(-l) % wf.shape[0]] - lines[l])
        assert diff.max() <= 2 and np.mean(diff != 0) < 0.02, (l, diff.max())
    assert not np.array_equal(res[0]["timf3"], res[1]["timf3"]) and np.count_nonzero(res[0]["timf3"]) > 0
    for r, key in ((0, "baseb_raw"), (1, "baseb_raw_orthog")):
        assert res[r]["baseb_pa"] == g["baseb_ptrs"][0]
        assert rel(res[r]["baseb"], g[key]) < 4e-6, (key, rel(res[r]["baseb"], g[key]))
