"""The code objects, checked without a GPU: every pipeline / lane-systolic instantiation keeps its state in registers (no
scratch), the binary16 tier's column loop has exactly its 4 x T v_pk_fma_f16, no v_perm_b32 and no 16-bit re-pack (round 3: the 20-, 24- and
28-row group-resident instantiations carried T shifts + T perms at every step, 5-8 % of their throughput), and the register
counts stay inside the occupancy class the planner's rate table was measured with."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa  # noqa: E402


@pytest.fixture(scope="module")
def isa():
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "swimm_amd", "csrc"), "isa"])
    return check_isa.kernels(os.path.join(ROOT, "swimm_amd", "csrc", "obj", "sw_kernels.s"))


def test_every_kernel_stays_in_registers(isa):
    assert len([k for k in isa if k["kernel"] == "sw_pipe_kernel"]) == 42 and len([k for k in isa if k["kernel"] == "sw_lane_kernel"]) == 9
    for k in isa:
        assert k["scratch_bytes"] == 0, k


def test_binary16_column_loop_has_no_repack(isa):
    seen = 0
    for k in isa:
        if k["kernel"] != "sw_pipe_kernel" or k["mode"] != 2:
            continue
        seen += 1
        assert k["v_pk_fma_f16"] == 4 * k["rows_per_wave"], k     # 4 columns per step, one fused pair-score + diagonal per row
        assert k["v_perm_b32"] == 0 and k["shifts_by_16"] == 0, k
    assert seen == 24        # 8 strip heights x (static, dynamic, dynamic group-resident)


def test_register_classes(isa):
    # waves per SIMD the launch shapes rely on (512 VGPRs per SIMD lane, allocated in eights): <= 128 -> 4 (16-wave workgroups),
    # <= 168 -> 3 (12-wave workgroups of the 32- and 36-row kernels); the lane-systolic kernel fits beside 3 x 144
    for k in isa:
        if k["kernel"] == "sw_pipe_kernel" and k["mode"] == 2:
            assert k["vgprs"] <= (128 if k["rows_per_wave"] <= 28 else 168), k
        if k["kernel"] == "sw_lane_kernel":
            assert k["vgprs"] <= 80, k
