"""The CU reservation of the factorisation (chol.hip: cu_blocker_kernel) works through REGISTER COUNTS alone: the blockers leave 216
vector registers per SIMD, kernels that need more cannot settle on the reserved CUs, kernels that need no more can.  A compiler that
allocates differently silently undoes it (it did once: the trapezoid kernel came out at 208 registers and leaked onto the reserved CUs).
This test compiles the two sources for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and checks the budget."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scikit-gpuppy_amd", "csrc")
FREE = 216   # 512 - 296 held by a blocker wave


def resource_usage(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o",
                          str(tmp_path / (src + ".o")), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    usage, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return usage


def alloc(u):
    """registers a wave of the kernel occupies: arch VGPRs in blocks of 8, plus its accumulation registers"""
    return (u["VGPRs"] + 7) // 8 * 8 + (u.get("AGPRs", 0) + 7) // 8 * 8


def pick(usage, fragment):
    hits = {k: v for k, v in usage.items() if fragment in k}
    assert hits, "no kernel matching %r among %s" % (fragment, sorted(usage))
    return hits


@pytest.mark.timeout(1800)
def test_register_budget_of_the_cu_reservation(tmp_path):
    chol = resource_usage("chol.hip", tmp_path)
    gemm = resource_usage("gemm.hip", tmp_path)
    for u in pick(chol, "cu_blocker_kernel").values():
        assert alloc(u) == 512 - FREE, u                      # what the blockers hold
    for u in pick(chol, "potrf_trtri128_elim_kernel").values():
        assert alloc(u) <= FREE, u                            # the leaf must fit beside a blocker
    # kernels that must stay OFF the reserved CUs: the 128 x 128-tile bulk kernels
    for frag in ("gemm_nt_f64_trap_signal_kernel", "gemm_nt_f64_kernelILi4ELi4E"):
        for k, u in pick(gemm, frag).items():
            assert alloc(u) > FREE, (k, u)
    # the chain's own small GEMMs must fit on them (two waves per SIMD)
    for frag in ("gemm_nt_f64_kernelILi2ELi2E", "gemm_nt_f64_kernelILi1ELi4E", "gemm_nt_f64_kernelILi1ELi1E"):
        for k, u in pick(gemm, frag).items():
            assert 2 * alloc(u) <= FREE, (k, u)


@pytest.mark.timeout(1800)
def test_dataflow_kernel_keeps_two_workgroups_per_cu(tmp_path):
    """The persistent dataflow kernel (dflow.hip) hides a workgroup's between-task work behind its CU mate's products: it needs TWO
    workgroups per CU -- at most 256 registers per wave (arch + accumulation: a granule of 32 AGPRs that nothing uses once pushed it to
    288 and to one workgroup per CU, silently) and at most 80 KB of LDS per workgroup."""
    usage = resource_usage("dflow.hip", tmp_path)
    for k, u in pick(usage, "chol_dataflow_kernel").items():
        assert alloc(u) <= 256, (k, u)
        assert u["LDS"] <= 80 * 1024, (k, u)
