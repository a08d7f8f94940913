"""Build-time guard on the generated gfx950 ISA of the headline kernel.

A 16-byte buffer store reads its data registers over two passes.  With a
register scalar offset hipcc (ROCm 7.2) scheduled the next VALU write of those
registers directly behind the store and gfx950 stored garbage in some lanes
(nsol_pdk.hip, bst()).  The kernel now uses a constant scalar offset, for which
the compiler keeps a wait state; this test compiles the file to assembly (no
GPU needed) and checks that no 16-byte buffer store is immediately followed by an
instruction that overwrites its data registers, and that none uses a register
scalar offset."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


_ASM = {}


def _sources_with_16_byte_buffer_stores():
    """Translation units that issue 16-byte buffer stores: in their own text, or
    through nsol_blur3_dma.hpp (whose kernels are compiled where
    NSOL_BLUR3_DMA_IMPL is defined)."""
    csrc = os.path.join(ROOT, "nsol_amd", "csrc")
    assert "raw_buffer_store_b128" in open(
        os.path.join(csrc, "nsol_blur3_dma.hpp")).read()
    out = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            text = open(os.path.join(csrc, f)).read()
            if "raw_buffer_store_b128" in text or \
                    "#define NSOL_BLUR3_DMA_IMPL" in text:
                out.append(f)
    return out


def _assembly(src, tmp_path_factory):
    """gfx950 assembly of one translation unit.  The first request compiles every
    guarded file side by side (the blur's instantiations take a minute and a
    half each)."""
    if src not in _ASM:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("hipcc not available")
        from concurrent.futures import ThreadPoolExecutor
        todo = sorted(set(_sources_with_16_byte_buffer_stores() + [src]) - set(_ASM))
        tmp = tmp_path_factory.mktemp("isa")

        def one(f):
            out = tmp / (f + ".s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17",
                            "-ffp-contract=off", "-I",
                            os.path.join(ROOT, "include"), "-S",
                            "--cuda-device-only", "-o", str(out),
                            os.path.join(ROOT, "nsol_amd", "csrc", f)],
                           check=True, stderr=subprocess.DEVNULL)
            return f, out.read_text()
        with ThreadPoolExecutor(max_workers=4) as pool:
            for f, text in pool.map(one, todo):
                _ASM[f] = text
    return _ASM[src]


@pytest.mark.parametrize("src,min_stores", [("nsol_pdk.hip", 100),
                                            ("nsol_conv.hip", 4),
                                            ("nsol_blur3_f32.hip", 100),
                                            ("nsol_blur3_f64.hip", 100),
                                            ("nsol_blur3_lz_f32.hip", 100),
                                            ("nsol_blur3_lz_f64.hip", 50)])
def test_no_store_data_hazard(src, min_stores, tmp_path_factory):
    """Every translation unit that issues 16-byte buffer stores (the headline
    kernel and the one-pass blur, config 4's A / A^T)."""
    assert src in _sources_with_16_byte_buffer_stores()
    text = _assembly(src, tmp_path_factory)
    ins = []
    for line in text.split("\n"):
        t = line.strip()
        if t and not t.startswith((";", ".")) and not t.endswith(":"):
            ins.append(t)
    stores = 0
    for i, t in enumerate(ins):
        if not t.startswith("buffer_store_dwordx4"):
            continue
        stores += 1
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        # vdata, vaddr, srsrc, soffset [modifiers]
        assert ops[3].split()[0] == "0", "register scalar offset: %s" % t
        data = _regs(ops[0])
        nxt = ins[i + 1] if i + 1 < len(ins) else ""
        if nxt.startswith(("v_", "ds_read", "buffer_load", "global_load")):
            dst = _regs(nxt.split(None, 1)[1].split(",")[0].strip())
            assert not (data & dst), "store data overwritten at once: %s | %s" % (
                t, nxt)
    assert stores >= min_stores, stores


def test_every_file_with_16_byte_buffer_stores_is_guarded():
    assert _sources_with_16_byte_buffer_stores() == [
        "nsol_blur3_f32.hip", "nsol_blur3_f64.hip", "nsol_blur3_lz_f32.hip",
        "nsol_blur3_lz_f64.hip", "nsol_conv.hip", "nsol_pdk.hip"]


def test_headline_instantiations_do_not_spill(tmp_path_factory):
    # the headline instantiation (float, depth 3, 12 waves, TV, l2, unit spacing)
    # must not spill: a few scratch dwords in its plane loop cost several percent
    text = _assembly("nsol_pdk.hip", tmp_path_factory)
    names = re.findall(r"\.name:\s+(\S+)", text)
    scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)
    head = [int(p) for n, p in zip(names, scratch)
            if "k_pd_fusedkIfLi4ELi12ELi3ELi3ELb0ELb0ELb0ELb1ELb0E" in n]
    assert head == [0], head
    # its counterpart for rows that are not a multiple of 16 bytes
    ragged = [int(p) for n, p in zip(names, scratch)
              if "k_pd_fusedkIfLi4ELi12ELi3ELi3ELb0ELb0ELb0ELb1ELb1E" in n]
    assert ragged == [0], ragged


def test_one_pass_blur_instantiations_do_not_spill(tmp_path_factory):
    # config 4's A / A^T (13 taps, 16 waves: 128 registers per lane) in all its
    # forms -- isotropic or not, with the LSMR epilogue or the Lanczos sums, ragged
    # rows
    for suf, t in (("f32", "fLi4E"), ("f64", "dLi2E")):
        # (the two halves of a Lanczos step are compiled in a unit of their own)
        text = _assembly("nsol_blur3_%s.hip" % suf, tmp_path_factory) + \
            _assembly("nsol_blur3_lz_%s.hip" % suf, tmp_path_factory)
        src = "nsol_blur3_[lz_]%s.hip" % suf
        names = re.findall(r"\.name:\s+(\S+)", text)
        scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)
        hot = [int(p) for n, p in zip(names, scratch)
               if "k_blur3_dmaI%sLi13ELi16E" % t in n]
        # (plain, epilogue, sums) x (isotropic or not) x (ragged or not); float: + the
        # two halves of a Lanczos step, the lean second half and the loss epilogue x
        # (isotropic or not)
        assert len(hot) == (20 if t[0] == "f" else 12) and not any(hot), (src, hot)
        # the Lanczos halves at every tap count they are built for (5 .. 13, double 5 .. 9)
        lz = {n: int(p) for n, p in zip(names, scratch)
              if re.search(r"k_blur3_dmaI%sLi\d+ELi16ELb[01]ELi[34]ELb0E" % t, n)}
        assert len(lz) == (5 if t[0] == "f" else 3) * 2 * 2 and not any(lz.values()), (src, lz)
        # the lean second half (EPI 6): float 5 .. 13 taps, double 5 .. 11
        l6 = {n: int(p) for n, p in zip(names, scratch)
              if re.search(r"k_blur3_dmaI%sLi\d+ELi16ELb[01]ELi6ELb0E" % t, n)}
        assert len(l6) == (5 if t[0] == "f" else 4) * 2 and not any(l6.values()), (src, l6)
        # the loss epilogue (EPI 5) at the same tap counts
        ls = {n: int(p) for n, p in zip(names, scratch)
              if re.search(r"k_blur3_dmaI%sLi\d+ELi16ELb[01]ELi5ELb0E" % t, n)}
        assert len(ls) == (5 if t[0] == "f" else 3) * 2 and not any(ls.values()), (src, ls)


def test_one_pass_outer_step_does_not_spill(tmp_path_factory):
    """k_admm_vw_g (nsol_ops.hip: ADMM's outer step and the next solve's start vector in
    one pass) carries a plane of state per lane: every instantiation must stay out of
    scratch memory, and the float32 form without b_reg -- BASELINE config 4's -- within the
    128 registers that keep four waves on a SIMD."""
    asm = _assembly("nsol_ops.hip", tmp_path_factory)
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*k_admm_vw_g\S*)(.*?)\.end_amdhsa_kernel",
                         asm, re.S):
        name, body = m.group(1), m.group(2)
        seen += 1
        scratch = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert scratch and int(scratch.group(1)) == 0, name
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert vg, name
        if "IfLi4ELb0E" in name:                    # <float, 4, false>
            assert int(vg.group(1)) <= 128, (name, vg.group(1))
    assert seen == 4
