"""Per-kernel registers / scratch / occupancy out of the compiler's resource reports (csrc/*.res): `python tools/res_summary.py [filter ...]`."""
import glob, os, re, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "easy_gaussian_splatting_amd", "csrc")
pat = re.compile(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
for f in sorted(glob.glob(os.path.join(here, "*.res"))):
    for m in pat.finditer(open(f).read()):
        if len(sys.argv) == 1 or any(k in m.group(1) for k in sys.argv[1:]):
            print(f"{m.group(1)[:72]:72s} sgpr {m.group(2):>3s} vgpr {m.group(3):>3s} scratch {m.group(4):>3s} waves {m.group(5)} lds {m.group(6)}")
