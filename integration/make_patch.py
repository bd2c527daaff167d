#!/usr/bin/env python3
"""Generates integration/linrad_hip.patch: the change set that makes liblinrad_hip.so selectable in Linrad as fft1 version 21.

Run in the build container (needs the reference tree, default /root/reference):  python3 integration/make_patch.py
The edits are anchored on single lines of the reference snapshot and are almost all pure insertions; the patch is written
with zero context lines (diff -U0), so it holds our added lines, their line numbers and the two one-line replacements --
no reference source is stored in this repository.  integration/check_patch.sh applies it to a scratch copy and compiles
the touched objects.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
INC = '#include "hipshim.h"\n'


def after_last_include(lines):
    idx = max(i for i, l in enumerate(lines[:120]) if l.startswith("#include"))
    lines.insert(idx + 1, INC)


def insert(lines, anchor, new, where="after", start=0, nth=1):
    """insert `new` (a string of whole lines) before/after the nth line at or behind `start` that matches the regex `anchor`"""
    n = 0
    for i in range(start, len(lines)):
        if re.search(anchor, lines[i]):
            n += 1
            if n == nth:
                pos = i + 1 if where == "after" else i
                lines[pos:pos] = new.splitlines(keepends=True)
                return i
    raise SystemExit(f"anchor not found: {anchor}")


def replace(lines, anchor, old, new):
    for i, l in enumerate(lines):
        if re.search(anchor, l):
            assert old in l, (anchor, l)
            lines[i] = l.replace(old, new)
            return
    raise SystemExit(f"anchor not found: {anchor}")


def func_top(lines, signature, new):
    """first statement of a function: right behind the opening brace that follows its signature line"""
    for i, l in enumerate(lines):
        if re.match(signature, l) and ";" not in l:          # the definition, not a prototype
            j = i
            while "{" not in lines[j]:
                j += 1
            lines[j + 1:j + 1] = new.splitlines(keepends=True)
            return
    raise SystemExit(f"function not found: {signature}")


def edit_globdef(L):
    insert(L, r"^#define GPU_CUDA 2", "#define GPU_HIP 3\n")
    replace(L, r"define MAX_FFT_VERSIONS", "21", "23")


def edit_fft1var(L):
    # fft_cntrl[21]: window storage 1, no permute table, max_n 16 (65536 with the second fft off; buf.c:335 caps it at 15 with it on), gpu = GPU_HIP
    insert(L, r'"Double precision"\}', ',{1,0,16,0,0,GPU_HIP,0,1,0,   "HIP MI355X"}                        //21\n'
           # real samples ("normal audio" / direct sampling): permute 2 like the reference's own real version 2, which is what make_filcorrstart
           # (fft1.c:4659) and the back-transform table (buf.c:1033, 1318) look at; the window storage mode is host-only and unused
           ',{1,2,14,0,0,GPU_HIP,0,1,0,   "HIP MI355X real"}                   //22\n')
    replace(L, r"1 chan direct conversion \(IQ\)", "19, -1}", "19, 21}")
    replace(L, r"1 chan normal audio", "4, -1,", "4, 22,")
    replace(L, r"2 chan normal audio", "4, -1,", "4, 22,")                    # two real channels per frame (fft1_reherm_dit_two, fft1_re.c:133-231)
    replace(L, r"2 chan direct conversion \(IQ\)", "20, -1}", "20, 21}")     # two RF channels: one context per channel behind the same hooks


def edit_buf(L):
    # get_wideband_sizes: version 21 takes the GPU sizing branches (batch = 2^gpu.fft1_batch_n blocks per fft1_b call, buf.c:248-257)
    insert(L, r"^if\(fft1_use_gpu\)\s*$", "if(fft_cntrl[FFT1_CURMODE].gpu == GPU_HIP)fft1_use_gpu=GPU_HIP;\n", where="before")


def edit_wcw(L):
    after_last_include(L)
    i = insert(L, r"^void wideband_dsp\(void\)", "", where="after")
    # context creation where the clFFT / cuFFT plans are made, before the worker threads start
    insert(L, r"ui\.network_flag & NET_RXIN_TIMF2\) != 0 &&", "if(fft1_use_gpu == GPU_HIP)\n  {\n  if(hip_open() != 0)\n    {\n    lirerr(1463);\n    goto errexit;\n    }\n  }\n",
           where="before", start=i)
    insert(L, r"^errexit:;", "if(fft1_use_gpu == GPU_HIP)hip_close();\n", start=i)
    func_top(L, r"^void compute_timf2_powersum\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_compute_timf2_powersum();return;}\n")
    # NET_RXOUT_FFT1 (wcw.c:1024-1043): the multicast payload is memcpy'd from the host ring fft1_float, which version 21 leaves empty: the
    # transforms of the batch just retired are brought into it first (worker path: the block worker k read; no-worker path: the block before timf1p_px)
    i = insert(L, r"^void wideband_dsp\(void\)", "", where="after")
    insert(L, r"memcpy\(&fft1_netsend_buffer\[fft1net_pa\],", "            if(fft1_use_gpu == GPU_HIP)hip_net_fft1(inptr_fft1b[k], fft1_pa);\n", where="before", start=i, nth=1)
    insert(L, r"memcpy\(&fft1_netsend_buffer\[fft1net_pa\],", "          if(fft1_use_gpu == GPU_HIP)hip_net_fft1((timf1p_px-timf1_blockbytes+timf1_bytes)&timf1_bytemask, fft1_pa);\n",
           where="before", start=i, nth=2)


def edit_fft1(L):
    after_last_include(L)
    i = insert(L, r"^void fft1_b\(", "", where="after")
    insert(L, r"^\s+default:\s*$", "    case 21:\n    case 22:\n// HIP on MI355X: the transform, the correction of fft1_c and every ring behind it stay on the device.\n"
           "    multiplicity=gpu_fft1_batch_size;\n    if(hip_fft1_b(timf1p_ref, out, gpu_handle_number) != 0)lirerr(1464);\n    goto fft_done;\n\n",
           where="before", start=i)
    # the switch of the two-channel branch (fft1.c:3686-3900): the same case -- hip_fft1_b runs one context per channel
    insert(L, r"^\s+default:\s*$", "    case 21:\n    case 22:\n    multiplicity=gpu_fft1_batch_size;\n    if(hip_fft1_b(timf1p_ref, out, gpu_handle_number) != 0)lirerr(1464);\n    goto fft_done;\n\n",
           where="before", start=i, nth=2)
    func_top(L, r"^void fft1_c\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_fft1_c();return;}\n")


def edit_timf2(L):
    after_last_include(L)
    func_top(L, r"^void make_timf2\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_make_timf2();return;}\n")


def edit_blank1(L):
    after_last_include(L)
    func_top(L, r"^void first_noise_blanker\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_first_noise_blanker();return;}\n")


def edit_fft2(L):
    after_last_include(L)
    func_top(L, r"^void make_fft2\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_make_fft2();return;}\n")


def edit_mix1(L):
    after_last_include(L)
    func_top(L, r"^void fft2_mix1_fixed\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_fft2_mix1_fixed();return;}\n")
    # the reference's default operating modes: second fft off (uivar.c:371-392 column 8) and AFC on (weak-signal CW)
    func_top(L, r"^void fft1_mix1_fixed\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_fft1_mix1_fixed();return;}\n")
    func_top(L, r"^void fft2_mix1_afc\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_fft2_mix1_afc();return;}\n")
    func_top(L, r"^void fft1_mix1_afc\(void\)", "if(fft1_use_gpu == GPU_HIP){hip_fft1_mix1_afc();return;}\n")


def edit_sellim(L):
    after_last_include(L)
    func_top(L, r"^void fft1_update_liminfo\(void\)", "if(fft1_use_gpu == GPU_HIP && hip_fft1_update_liminfo())return;\n")
    func_top(L, r"^void fft2_update_liminfo\(void\)", "if(fft1_use_gpu == GPU_HIP && hip_fft2_update_liminfo())return;\n")


def edit_spursub(L):
    after_last_include(L)
    # acquisition of a carrier (init_spur_elimination's two calls, spursub.c:304-309) on the device-resident spectra; the subtraction from
    # the transforms already in the ring is skipped (they are on the device; the loop subtracts from the next transform on)
    func_top(L, r"^int store_new_spur\(int pnt\)", "if(fft1_use_gpu == GPU_HIP)return hip_store_new_spur(pnt);\n")
    func_top(L, r"^int spur_phase_lock\(int nx\)", "if(fft1_use_gpu == GPU_HIP)return hip_spur_phase_lock(nx);\n")
    func_top(L, r"^void initial_remove_spur\(void\)", "if(fft1_use_gpu == GPU_HIP)return;\n")
    func_top(L, r"^void swap_spurs\(int ia, int ib\)", "if(fft1_use_gpu == GPU_HIP){hip_swap_spurs(ia,ib);return;}\n")


def edit_spur(L):
    after_last_include(L)
    func_top(L, r"^void remove_spur\(int ia\)", "if(fft1_use_gpu == GPU_HIP){hip_remove_spur(ia);return;}\n")


def edit_rxin(L):
    after_last_include(L)
    i = insert(L, r"^void finish_rx_read\(", "", where="after")
    # the block the input thread has just filled sits at timf1_char[timf1p_pa]: hand it to the device before the event is posted
    insert(L, r"^// Set the EVENT_TIMF1 condition", "if(fft1_use_gpu == GPU_HIP)hip_timf1_new(timf1p_pa, snd[RXAD].block_bytes);\n", where="before", start=i)
    # NET_RXOUT_TIMF2 / NET_RXOUT_FFT2 (rxin.c:919-1060): the sender walks the host rings timf2_float / fft2_float from timf2_pt / fft2_pt;
    # with version 21 the span it is about to walk is fetched from the device first
    insert(L, r"net_rxdata_timf2\.userx_no=-ui\.rx_rf_channels;", "      if(fft1_use_gpu == GPU_HIP)hip_net_timf2(timf2_pt, mm);\n", where="before")
    insert(L, r"charbuf=\(char\*\)\(fft2_float\);", "        if(fft1_use_gpu == GPU_HIP)hip_net_fft2(fft2_pt, mm/4);\n", where="before")


EDITS = {"globdef.h": edit_globdef, "fft1var.c": edit_fft1var, "buf.c": edit_buf, "wcw.c": edit_wcw, "fft1.c": edit_fft1,
         "timf2.c": edit_timf2, "blank1.c": edit_blank1, "fft2.c": edit_fft2, "mix1.c": edit_mix1, "sellim.c": edit_sellim, "rxin.c": edit_rxin,
         "spursub.c": edit_spursub, "spur.c": edit_spur}


def main():
    out = []
    with tempfile.TemporaryDirectory() as td:
        for name, fn in EDITS.items():
            src = os.path.join(REF, name)
            L = open(src, encoding="latin-1").read().splitlines(keepends=True)
            fn(L)
            dst = os.path.join(td, name)
            open(dst, "w", encoding="latin-1").write("".join(L))
            r = subprocess.run(["diff", "-U0", "--label", "a/" + name, "--label", "b/" + name, src, dst], stdout=subprocess.PIPE, text=True, encoding="latin-1")
            assert r.returncode == 1, (name, r.returncode)
            out.append(r.stdout)
    open(os.path.join(HERE, "linrad_hip.patch"), "w", encoding="latin-1").write("".join(out))
    print("wrote", os.path.join(HERE, "linrad_hip.patch"), sum(o.count("\n") for o in out), "lines")


if __name__ == "__main__":
    main()
