"""Register / LDS / scratch figures of the kernels inside llm_amd/libggml_hip.so: extracts the embedded gfx950 code object and
prints the metadata notes of the kernels whose mangled name contains the given substrings (all of them if none given).
    python tests/tools/kernel_regs.py k_qkv_attn k_mmvq_big"""
import os
import re
import subprocess
import sys
import tempfile

so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm_amd", "libggml_hip.so")
data = open(so, "rb").read()
co = None
for m in re.finditer(b"\x7fELF", data):
    i = m.start()
    if i and int.from_bytes(data[i + 18:i + 20], "little") == 0xE0:  # EM_AMDGPU
        shoff = int.from_bytes(data[i + 0x28:i + 0x30], "little")
        size = shoff + int.from_bytes(data[i + 0x3A:i + 0x3C], "little") * int.from_bytes(data[i + 0x3C:i + 0x3E], "little")
        co = data[i:i + size]
        break
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
cur = {}
rows = []
for line in notes.splitlines():
    mm = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not mm:
        continue
    k, v = mm.groups()
    if k == "agpr_count" and cur:
        rows.append(cur)
        cur = {}
    cur[k] = v
if cur:
    rows.append(cur)
want = sys.argv[1:]
print(f"{'vgpr':>5} {'sgpr':>5} {'spill':>6} {'scratch':>8} {'lds':>7}  kernel")
for r in rows:
    name = r.get("name", "")
    if want and not any(w in name for w in want):
        continue
    print(f"{r.get('vgpr_count', '?'):>5} {r.get('sgpr_count', '?'):>5} {r.get('vgpr_spill_count', '?'):>6} "
          f"{r.get('private_segment_fixed_size', '?'):>8} {r.get('group_segment_fixed_size', '?'):>7}  {name[:110]}")
