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
    assert len([k for k in isa if k["kernel"] == "sw_pipe_kernel"]) == 50 and len([k for k in isa if k["kernel"] == "sw_lane_kernel"]) == 9
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
    assert seen == 32        # 8 strip heights x (static, dynamic, dynamic group-resident, dynamic over a list that is still landing)


def test_register_classes(isa):
    # waves per SIMD the launch shapes rely on (512 VGPRs per SIMD lane, allocated in eights): <= 128 -> 4 (16-wave workgroups),
    # <= 168 -> 3 (12-wave workgroups of the 32- and 36-row kernels); the lane-systolic kernel fits beside 3 x 144
    for k in isa:
        if k["kernel"] == "sw_pipe_kernel" and k["mode"] == 2 and not k["growing_list"]:
            assert k["vgprs"] <= (128 if k["rows_per_wave"] <= 28 else 168), k
        # The instantiations that walk a growing item list wait, resident, while the upload stream's tiling kernels must run on the
        # SAME CUs (workgroups are dealt to shader engines before a free CU is looked for: tools/microbench/spin_probe): the host
        # only launches them in shapes that leave a tiling wave its 32 registers on every SIMD (search.cpp, layout_ranges), reading
        # the count from the code object -- here: every strip height has such a shape (4 waves x 120, or fewer waves)
        if k["kernel"] == "sw_pipe_kernel" and k["growing_list"]:
            assert k["vgprs"] <= 160 and (k["vgprs"] <= 120 or k["rows_per_wave"] >= 24), k
        # the tiling kernels and the publisher run while pipeline waves fill the chip: they must fit what 4 x 120 registers leave of 512
        if k["kernel"] in ("retile_kernel", "retile4_kernel", "tile_sequences_kernel", "publish_items_kernel"):
            assert k["vgprs"] <= 32 and k["scratch_bytes"] == 0, k
        if k["kernel"] == "sw_lane_kernel":
            assert k["vgprs"] <= 80, k
    assert {"retile_kernel", "retile4_kernel", "tile_sequences_kernel", "publish_items_kernel"} <= {k["kernel"] for k in isa}
