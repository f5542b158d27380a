"""Disassemble the gfx950 code objects embedded in a HIP shared library / object and count instructions by pattern.

Used by tests/test_abi.py as a build guard: libstk.so must not contain packed-fp32 VALU instructions (v_pk_add_f32 /
v_pk_mul_f32 / v_pk_fma_f32 / v_pk_mov_b32).  On gfx950 such an instruction whose op_sel makes one half of the result read the
OTHER register of a 64-bit source pair returns a wrong value in lanes 48..63 while a wave of another kernel issues MFMAs on
the same SIMD (DESIGN.md "The hazard"; tools/_probe/cores2.hip is the stand-alone reproducer).

  python tools/isa_scan.py soft-truncation_amd/csrc/libstk.so [regex] [--fail] [--arch=gfx950]

--fail: exit status 1 when anything matches OR when nothing was scanned (no code object of that architecture found, or an empty
disassembly: a bundle-format / objdump mismatch must not pass as "no hits").
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PACKED_F32 = r'\bv_pk_(add|mul|fma)_f32\b|\bv_pk_mov_b32\b'


def code_objects(path, arch='gfx950'):
  """The device code objects (bytes) for `arch` of every offload bundle found in the file."""
  blob = open(path, 'rb').read()
  out = []
  pos = blob.find(MAGIC)
  while pos >= 0:
    n, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
    p = pos + len(MAGIC) + 8
    for _ in range(n):
      off, size, tlen = struct.unpack_from('<QQQ', blob, p)
      triple = blob[p + 24:p + 24 + tlen].decode()
      p += 24 + tlen
      if arch in triple and size:
        out.append(blob[pos + off:pos + off + size])
    pos = blob.find(MAGIC, pos + len(MAGIC))
  return out


def disassemble(path, arch='gfx950'):
  """Yield the disassembly text of every device code object in `path`."""
  for co in code_objects(path, arch):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
      f.write(co)
      f.flush()
      yield subprocess.run([OBJDUMP, '-d', f'--mcpu={arch}', f.name], capture_output=True, text=True, check=True).stdout


def scan(path, pattern=PACKED_F32, arch='gfx950'):
  """(number of code objects, number of instructions, [(kernel symbol, instruction text)] matching `pattern`)."""
  rx = re.compile(pattern)
  hits = []
  n_obj = n_inst = 0
  for text in disassemble(path, arch):
    n_obj += 1
    sym = '?'
    for line in text.splitlines():
      m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
      if m:
        sym = m.group(1)
        continue
      if '//' in line:
        n_inst += 1
        if rx.search(line):
          hits.append((sym, line.split('//')[0].strip()))
  return n_obj, n_inst, hits


if __name__ == '__main__':
  fail = '--fail' in sys.argv            # exit status 1 when anything matches or nothing was scanned (the Makefile's post-link guard)
  arch = ([a.split('=', 1)[1] for a in sys.argv if a.startswith('--arch=')] or ['gfx950'])[-1]
  argv = [a for a in sys.argv if a != '--fail' and not a.startswith('--arch=')]
  path = argv[1]
  pattern = argv[2] if len(argv) > 2 else PACKED_F32
  n_obj, n_inst, hits = scan(path, pattern, arch)
  print(f'{path}: {n_obj} {arch} code objects, {n_inst} instructions, {len(hits)} matching /{pattern}/')
  per = {}
  for sym, inst in hits:
    per.setdefault(sym, []).append(inst)
  for sym, lst in sorted(per.items(), key=lambda kv: -len(kv[1]))[:40]:
    print(f'  {len(lst):5d}  {sym}    e.g. {lst[0]}')
  if fail and (hits or n_obj == 0 or n_inst == 0):
    if not hits:
      print(f'{path}: nothing scanned for {arch} -- refusing to call that clean')
    sys.exit(1)
