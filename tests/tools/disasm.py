"""Disassembly of one kernel inside llm_amd/libggml_hip.so (the embedded gfx950 code object):
    python tests/tools/disasm.py <substring of the mangled name> [out.s]"""
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
want = sys.argv[1]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "-W", f.name], capture_output=True, text=True).stdout
    names = [l.split()[-1] for l in syms.splitlines() if want in l and " FUNC " in l]
    if not names:
        sys.exit("no kernel matches " + want)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", "--disassemble-symbols=" + names[0], f.name],
                         capture_output=True, text=True).stdout
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
    print(names[0], len(out.splitlines()), "lines ->", sys.argv[2])
else:
    print(out)
