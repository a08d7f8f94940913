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


def test_no_store_data_hazard_in_k_pd_fusedk(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "pdk.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17",
                    "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "nsol_amd", "csrc", "nsol_pdk.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    ins = []
    for line in out.read_text().split("\n"):
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
    assert stores > 100
    # the headline instantiation (float, depth 3, 12 waves, TV, l2, unit spacing)
    # must not spill: a few scratch dwords in its plane loop cost several percent
    text = out.read_text()
    names = re.findall(r"\.name:\s+(\S+)", text)
    scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)
    head = [int(p) for n, p in zip(names, scratch)
            if "k_pd_fusedkIfLi4ELi12ELi3ELi3ELb0ELb0ELb0ELb1ELb0E" in n]
    assert head == [0], head
    # its counterpart for rows that are not a multiple of 16 bytes
    ragged = [int(p) for n, p in zip(names, scratch)
              if "k_pd_fusedkIfLi4ELi12ELi3ELi3ELb0ELb0ELb0ELb1ELb1E" in n]
    assert ragged == [0], ragged
