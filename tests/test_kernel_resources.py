"""CPU: what the compiler made of the two persistent-wavefront kernels (the samplers' exact chains).  They run as ONE wavefront
each, at the speed of the waits and memory round trips in their instruction stream, so two properties of the generated code are
pinned here (round 6: `W.kk[q][j]`, a register array indexed by the walk's choice, had moved to scratch memory -- a memory round
trip per OR node of forest-em's chain, `ScratchSize: 112` -- and nothing in the source said so):
  * no scratch memory (no spills, no stack objects),
  * the lattice chain's backward sweep waits ONCE a level (its loop holds one `s_waitcnt lgkmcnt(0)`, for the values' read)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "carmel_amd", "csrc")


def device_asm(name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off",
                        "--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", "-"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def kernels(asm):
    """mangled kernel name -> its text (label to s_endpgm) and its resource comment block"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\ts_endpgm(.*?)(?=^_Z\w+:|\Z)", asm, flags=re.S | re.M):
        out[m.group(1)] = (m.group(2), m.group(3))
    return out


@pytest.mark.parametrize("src,needle", [("forest_exact.hip", "forest_exact_kernel"), ("gibbs_exact.hip", "gibbs_exact_wave_kernelILb0E")])
def test_the_exact_chains_use_no_scratch_memory(src, needle):
    ks = {k: v for k, v in kernels(device_asm(src)).items() if needle in k}
    assert ks, needle
    for name, (body, tail) in ks.items():
        m = re.search(r"; ScratchSize: (\d+)", tail)
        assert m, name
        assert int(m.group(1)) == 0, (name, m.group(0))
        assert "scratch_" not in body, name


def test_lattice_chain_backward_sweep_waits_once_a_level():
    """the level loop of gibbs_exact_wave_kernel<false> is ONE basic block that branches back to itself (no branch inside: every
    read is unconditional, idle lanes add zero to a slot of their own) with the LDS add and a single s_waitcnt in it"""
    ks = {k: v for k, v in kernels(device_asm("gibbs_exact.hip")).items() if "gibbs_exact_wave_kernelILb0E" in k}
    (body, _), = ks.values()
    blocks = re.split(r"^(\.LBB\d+_\d+):.*$", body, flags=re.M)  # [pre, label, text, label, text, ...]
    found = []
    for label, text in zip(blocks[1::2], blocks[2::2]):
        code = [ln for ln in text.splitlines() if ln.strip() and not ln.strip().startswith(";")]
        if not any("ds_add_f64" in ln for ln in code):
            continue
        branches = [ln for ln in code if "s_cbranch" in ln or "s_branch" in ln]
        if len(branches) == 1 and branches[0].split()[-1] == label and code[-1] == branches[0]:
            found.append(sum("s_waitcnt" in ln for ln in code))
    assert found == [1], found
