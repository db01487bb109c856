"""print (vgpr, spilled vgprs, scratch bytes, sgpr, LDS) of every kernel in a hipcc -save-temps device .s file whose name
contains the filter:  python tools/isa_regs.py /tmp/x-hip-amdgcn-amd-amdhsa-gfx950.s conv_nhwc"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()[:90]
    except Exception:
        pass
    print(f"agpr {blk.split()[0]:>4} vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} "
          f"sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6}  {name}")
