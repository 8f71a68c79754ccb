"""Development tool: disassemble the hiprtc-specialised kernels of one model.
    python tools/disasm.py K M DS [batchsize fantasy_hidden_len] > out.s
Compiles into a scratch cache directory (no GPU needed) and runs llvm-objdump on the code object;
a per-kernel resource summary (registers, spills, LDS) goes to stderr."""
import ctypes, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K, M, DS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
LF = int(sys.argv[5]) if len(sys.argv) > 5 else 200
d = tempfile.mkdtemp(prefix="crbm_disasm_")
os.environ["CRBM_JIT_CACHE"] = d
from crbm_amd.csrc import build
build.precompile([dict(num_motifs=K, motif_length=M, doublestranded=DS, batchsize=B, fantasy_hidden_len=LF)], verbose=False)
f = glob.glob(os.path.join(d, "*.hsaco"))[0]
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
cur = {}
for line in notes.splitlines():
    line = line.strip()
    for key in (".name:", ".vgpr_count:", ".agpr_count:", ".sgpr_count:", ".vgpr_spill_count:", ".sgpr_spill_count:", ".group_segment_fixed_size:", ".private_segment_fixed_size:"):
        if line.startswith(key) or line.startswith("- " + key):
            cur[key] = line.split(":", 1)[1].strip()
    if line.startswith(".wavefront_size") or line.startswith("- .wavefront_size"):
        pass
    if ".vgpr_spill_count:" in line and ".name:" in cur:
        sys.stderr.write("%-28s vgpr %s agpr %s sgpr %s spills v%s s%s scratch %s B\n" % (cur.get(".name:"), cur.get(".vgpr_count:"), cur.get(".agpr_count:"), cur.get(".sgpr_count:"), cur.get(".vgpr_spill_count:"), cur.get(".sgpr_spill_count:"), cur.get(".private_segment_fixed_size:")))
        cur = {}
print(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f], capture_output=True, text=True).stdout)
