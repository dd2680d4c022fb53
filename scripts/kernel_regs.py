"""Register / spill / LDS summary of every kernel in a hipcc -save-temps .s file (amdhsa metadata)."""
import re, sys, yaml
txt = open(sys.argv[1]).read()
m = re.search(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", txt, re.S)
meta = yaml.safe_load(m.group(1))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in meta["amdhsa.kernels"]:
    if pat in k[".name"]:
        print("%-90s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %4d lds %6d" % (
            k[".name"][-90:], k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"], k[".vgpr_spill_count"],
            k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))
