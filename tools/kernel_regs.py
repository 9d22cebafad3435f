"""Register / spill / scratch use of the kernels in one object of csrc/obj (metadata notes of the gfx950 code object):
   python tools/kernel_regs.py skm_gmm [name-substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "speaker-recognition_amd", "csrc", "obj", sys.argv[1] + ".o")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    if "--asm" in sys.argv:
        sys.stdout.write(subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True))
        sys.exit(0)
for k in re.split(r"\n  - \.agpr_count", notes)[1:]:
    name = subprocess.check_output(["c++filt", re.search(r"\.name:\s+(\S+)", k).group(1)], text=True).strip()
    if sys.argv[2:] and not any(a in name for a in sys.argv[2:]):
        continue
    f = lambda key: re.search(r"\.%s:\s+(\d+)" % key, k).group(1)
    print("%-70s agpr %3s vgpr %3s sgpr %3s spill %3s scratch %5s lds %6s" % (
        name[:70], re.match(r":\s+(\d+)", k).group(1), f("vgpr_count"), f("sgpr_count"), f("vgpr_spill_count"),
        f("private_segment_fixed_size"), f("group_segment_fixed_size")))
