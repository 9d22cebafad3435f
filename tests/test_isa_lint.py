"""Static lint of the generated gfx950 ISA (tools/isa_lint.py): clean on the shipped library, red on a build with one
hand-placed wait removed.  Needs hipcc and llvm-objdump (the build container), no GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIB = os.path.join(ROOT, "speaker-recognition_amd", "libskm.so")
needs_tools = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump"))),
                                 reason="hipcc / llvm-objdump not installed")


def _ins(lines):
    return [isa_lint.Ins(op, args, 0x1000 + 4 * k) for k, (op, args) in enumerate(lines)]


def test_rules_on_hand_written_sequences():
    mfma = ("v_mfma_f32_32x32x2_f32", "v[0:15], v20, v21, v[0:15]")
    # (i) a VALU read 3 wait states behind the MFMA; fine behind 18
    bad = _ins([mfma, ("s_nop", "1"), ("v_mov_b32_e32", "v30, v3"), ("s_endpgm", "")])
    assert any(f.startswith("(i)") for f in isa_lint.lint_kernel("k", bad))
    good = _ins([mfma, ("s_nop", "15"), ("s_nop", "1"), ("v_mov_b32_e32", "v30, v3"), ("s_endpgm", "")])
    assert isa_lint.lint_kernel("k", good) == []
    # the accumulation chain needs no wait; a second MFMA of the kind makes the first one 16 wait states old
    chain = _ins([mfma, mfma, ("v_mfma_f32_32x32x2_f32", "v[32:47], v20, v21, v[32:47]"), ("s_nop", "0"),
                  ("v_mov_b32_e32", "v30, v3"), ("s_endpgm", "")])
    assert isa_lint.lint_kernel("k", chain) == []
    isa_lint.STRICT = True          # LLVM's counting: one wait state per instruction
    try:
        assert any(f.startswith("(i)") for f in isa_lint.lint_kernel("k", chain))
    finally:
        isa_lint.STRICT = False
    # (i) through the taken arm of a branch
    taken = _ins([mfma, ("s_cbranch_vccnz", "2"), ("s_nop", "15"), ("s_nop", "3"), ("v_mov_b32_e32", "v30, v3"),
                  ("s_endpgm", "")])
    assert any(f.startswith("(i)") for f in isa_lint.lint_kernel("k", taken))
    # (ii) VALU write one wait state ahead of the MFMA that reads it
    bad = _ins([("v_mov_b32_e32", "v3, 0"), ("s_nop", "0"), mfma, ("s_endpgm", "")])
    assert any(f.startswith("(ii)") for f in isa_lint.lint_kernel("k", bad))
    good = _ins([("v_mov_b32_e32", "v3, 0"), ("s_nop", "1"), mfma, ("s_endpgm", "")])
    assert not any(f.startswith("(ii)") for f in isa_lint.lint_kernel("k", good))
    # (iii) a loop whose LDS-direct load is read in the next trip without a vmcnt wait, and with one
    loop = [("ds_read_b128", "v[4:7], v40"), ("buffer_load_dword", "v41, s[16:19], 0 offen lds"),
            ("s_cbranch_scc1", str(0x10000 - 3)), ("s_endpgm", "")]
    assert any(f.startswith("(iii)") for f in isa_lint.lint_kernel("k", _ins(loop)))
    loop = [("s_waitcnt", "vmcnt(0)")] + loop
    loop[3] = ("s_cbranch_scc1", str(0x10000 - 4))
    assert isa_lint.lint_kernel("k", _ins(loop)) == []


@needs_tools
def test_library_is_lint_clean():
    assert os.path.exists(LIB), "libskm.so is not built (speaker-recognition_amd/csrc/build.sh)"
    findings = isa_lint.lint_file(LIB)
    assert findings == [], "tools/isa_lint.py: %d kernels with findings" % len(findings)


@needs_tools
@pytest.mark.parametrize("macro, rule", [(None, None), ("SKM_LINT_DEMO_ODD_NO_SETTLE", "(i)"),
                                         ("SKM_LINT_DEMO_TAIL_NO_PAD", "(ii)"), ("SKM_LINT_DEMO_NO_DMA_WAIT", "(iii)")])
def test_lint_is_red_with_a_wait_removed(tmp_path, macro, rule):
    obj = str(tmp_path / "lint_demo.o")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(ROOT, "tools", "lint_demo.hip"),
           "-o", obj] + (["-D" + macro] if macro else [])
    subprocess.run(cmd, check=True, cwd=str(tmp_path))
    findings = isa_lint.lint_file(obj)
    if macro is None:
        assert findings == []
    else:
        lines = [line for _, f in findings for line in f]
        assert lines and all(line.startswith(rule) for line in lines), lines[:4]
        if macro == "SKM_LINT_DEMO_ODD_NO_SETTLE":      # round 2's wrong-logits bug, in the kernel that had it
            assert any("k_estep_solo" in isa_lint.demangle(name) for name, _ in findings)


@needs_tools
def test_lint_is_red_without_the_ring_wait_of_the_split_gemm(tmp_path):
    """csrc/skm_gemm_b3.hip keeps its LDS-direct loads in flight for two iterations (RING_DEPTH in the lint): built
    without the vmcnt(6) that retires a stage, rule (iii) must name every one of its loads."""
    obj = str(tmp_path / "b3.o")
    src = os.path.join(ROOT, "speaker-recognition_amd", "csrc", "skm_gemm_b3.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-DSKM_LINT_DEMO_B3_NO_DMA_WAIT", "-c", src, "-o", obj], check=True, cwd=str(tmp_path))
    findings = isa_lint.lint_file(obj)
    lines = [line for name, f in findings for line in f if "k_gemm_b3" in name]
    assert lines and all(line.startswith("(iii)") for line in lines), lines[:4]
    assert all("k_gemm_b3" in name for name, _ in findings)
