#!/usr/bin/env python3
"""Extract the Sobol' generator tables used by the reference sampler into a flat binary.

Reads /root/reference/src/pathtracer/sobolmatrices.rs *as text* (it is a pure data table:
SOBOL_MATRICES_32 at :7, VD_C_SOBOL_MATRICES at :53463, VD_C_SOBOL_MATRICES_INV at :54155)
and writes data/sobol_tables.bin.  Only numbers are taken; no reference code is copied.

Layout (little endian):
  char[8]  magic "PTRSSOB1"
  u32      num_dimensions (1024)
  u32      matrix_size    (52)
  u32      vdc_rows (25), u32 vdc_inv_rows (26), u32 row_stride (52), u32 reserved
  u32[1024*52]   SOBOL_MATRICES_32
  u32[25]        VDC row lengths,   u32[26] VDC_INV row lengths, u32 pad (8-byte align)
  u64[25*52]     VDC rows, zero padded to row_stride
  u64[26*52]     VDC_INV rows, zero padded to row_stride
"""
import re
import struct
import sys
from pathlib import Path

SRC = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/pathtracer/sobolmatrices.rs")
DST = Path(sys.argv[2] if len(sys.argv) > 2 else Path(__file__).resolve().parents[1] / "data" / "sobol_tables.bin")

text = SRC.read_text()
num_re = re.compile(r"0x([0-9a-fA-F_]+?)(?:_u64|_u32)?(?=[,\s\]])")


def parse_array(name):
    m = re.search(r"const\s+" + name + r"\s*:\s*\[[^\]]*\]\s*=\s*\[(.*?)\];", text, re.S)
    assert m, name
    return [int(h.replace("_", ""), 16) for h in num_re.findall(m.group(1) + " ")]


mats = parse_array("SOBOL_MATRICES_32")
assert len(mats) == 1024 * 52, len(mats)
vdc = [parse_array("M%d" % i) for i in range(1, 26)]
vdc_inv = [parse_array("MI%d" % i) for i in range(1, 27)]
# identities stated by the table's own declared lengths (sobolmatrices.rs:53257-54154)
for m, row in enumerate(vdc, start=1):
    assert len(row) == 2 * (26 - m), (m, len(row))
for m, row in enumerate(vdc_inv, start=1):
    assert len(row) == 2 * m, (m, len(row))
# dimension 0 is the bit-reversal (van der Corput) matrix
assert mats[:32] == [1 << (31 - i) for i in range(32)] and all(v == 0 for v in mats[32:52])

STRIDE = 52
out = bytearray()
out += b"PTRSSOB1"
out += struct.pack("<6I", 1024, 52, 25, 26, STRIDE, 0)
out += struct.pack("<%dI" % len(mats), *mats)
out += struct.pack("<25I", *[len(r) for r in vdc])
out += struct.pack("<26I", *[len(r) for r in vdc_inv])
out += struct.pack("<I", 0)
for row in vdc:
    out += struct.pack("<%dQ" % STRIDE, *(row + [0] * (STRIDE - len(row))))
for row in vdc_inv:
    out += struct.pack("<%dQ" % STRIDE, *(row + [0] * (STRIDE - len(row))))
DST.parent.mkdir(parents=True, exist_ok=True)
DST.write_bytes(bytes(out))
print("wrote", DST, len(out), "bytes")
