#!/usr/bin/env python3
"""Device ISA of the product library, read for the two things a reader of C++ cannot see:

  * 24-bit integer divisions. This toolchain expands `x / n` and `x % n` of operands it can prove to fit 24 bits through ONE float reciprocal
    (v_cvt_f32_u32 x2, v_rcp_iflag_f32, v_mul_f32, v_trunc_f32, v_fma_f32, v_cmp_ge_f32 |r|, n) - and that expansion returned 0xffffff for 2.8 % of the
    dividends at n = 11 on gfx950 (tools/urem_probe.hip, profiles/r04_step8_urem24_probe.txt): round 4's region server faulted on it. The 32-bit expansion
    (v_rcp_iflag_f32, v_mul_f32 0x4f7ffffe, v_mul_hi_u32 refinement) and the 64-bit one are exact. `divisions()` tells them apart.
  * register spills per kernel (.vgpr_spill_count / .sgpr_spill_count / scratch bytes of the kernel descriptors' metadata).

  python tools/isa_report.py [file.hip ...]      (default: octopus_amd/csrc/oct_phmm.hip; the assembly is cached by source digest under /tmp)
"""
from __future__ import annotations

import hashlib
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value"]


def device_asm(src: Path, cache_dir: Path = Path("/tmp/oct_phmm_isa")) -> str:
    """hipcc -S --cuda-device-only of one translation unit; cached by the digest of everything it can include from the tree."""
    src = Path(src)
    deps = [src] + sorted(src.parent.glob("*.hpp")) + sorted(src.parent.glob("*.hh")) + sorted((ROOT / "include").glob("*.h"))
    m = hashlib.sha256()
    for f in deps:
        m.update(f.name.encode()); m.update(f.read_bytes())
    m.update(" ".join(FLAGS).encode())
    cache_dir.mkdir(parents=True, exist_ok=True)
    out = cache_dir / f"{src.stem}-{m.hexdigest()[:16]}.s"
    if not out.exists():
        tmp = out.with_suffix(".tmp.s")
        subprocess.run([HIPCC] + FLAGS + [f"-I{ROOT / 'include'}", str(src), "-o", str(tmp)], check=True, capture_output=True)
        tmp.rename(out)
    return out.read_text()


def _kernel_bodies(asm: str):
    """(mangled name, [instruction lines]) per function of the assembly."""
    cur, body = None, []
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if cur is not None:
                yield cur, body
            cur, body = m.group(1), []
            continue
        t = line.split(";")[0].strip()
        if cur is not None and t and not t.startswith("."):
            body.append(t)
        elif cur is not None and t.startswith(".L") and t.endswith(":"):
            body.append(t)
    if cur is not None:
        yield cur, body


def divisions(asm: str):
    """Every float-reciprocal integer division expansion: (kernel, index, kind), kind in {"u24", "u32", "other"}.
    u24 is the one that must not exist: the quotient estimate is truncated and corrected ONCE by a float compare, with no integer refinement."""
    found = []
    for name, body in _kernel_bodies(asm):
        for i, ins in enumerate(body):
            if not ins.startswith("v_rcp_iflag_f32"):
                continue
            window = body[i + 1:i + 28]
            is32 = any("0x4f7ffffe" in w for w in window[:10])
            is24 = (not is32 and any(w.startswith("v_trunc_f32") for w in window)
                    and any(w.startswith(("v_cmp_ge_f32", "v_cmp_le_f32", "v_cmp_gt_f32", "v_cmp_lt_f32")) and "|" in w for w in window))
            found.append((name, i, "u32" if is32 else "u24" if is24 else "other"))
    return found


def spills(asm: str):
    """{kernel: (vgpr_spills, sgpr_spills, private_segment_bytes)} from the code-object metadata."""
    out = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", asm)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        g = lambda key: int((re.search(rf"\.{key}:\s+(\d+)", blk) or [0, 0])[1])
        out[nm.group(1)] = (g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"))
    return out


def spill_sites(asm: str):
    """{kernel: (scratch instructions, size of the SMALLEST loop that encloses any of them or 0, instructions in the kernel)} for the kernels that touch scratch memory:
    does a spill sit in a hot inner loop (every DP iteration pays a memory round trip) or only in the loop over task groups (once per group of 10^2 - 10^4 iterations)?"""
    out = {}
    for name, body in _kernel_bodies(asm):
        sites = [i for i, ins in enumerate(body) if ins.startswith("scratch_")]
        if not sites:
            continue
        labels = {ins[:-1]: i for i, ins in enumerate(body) if ins.endswith(":")}
        loops = []
        for i, ins in enumerate(body):
            m = re.match(r"s_cbranch_\w+ (\S+)", ins) or re.match(r"s_branch (\S+)", ins)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        enclosing = [b - a for i in sites for a, b in loops if a <= i <= b]
        out[name] = (len(sites), min(enclosing) if enclosing else 0, len(body))
    return out


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.split("\n")))
    except Exception:
        return {n: n for n in names}


def main(argv):
    srcs = [Path(a) for a in argv] or [ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip"]
    rc = 0
    for src in srcs:
        asm = device_asm(src)
        div = divisions(asm)
        sp = spills(asm)
        bad = [d for d in div if d[2] == "u24"]
        print(f"{src}: {len(sp)} kernels; integer divisions through a float reciprocal: {len(div)} "
              f"({sum(d[2] == 'u32' for d in div)} exact 32-bit, {sum(d[2] == 'other' for d in div)} other, {len(bad)} 24-bit)")
        spilled = {k: v for k, v in sp.items() if v[0] or v[1]}
        names = demangle(sorted(set([b[0] for b in bad] + list(spilled))))
        for name, i, _ in bad:
            print(f"  24-BIT DIVISION in {names[name][:160]} at instruction {i}")
            rc = 1
        sites = spill_sites(asm)
        for k, (v, s, scratch) in sorted(spilled.items()):
            where = (f"; {sites[k][0]} scratch instructions, smallest loop around any of them: {sites[k][1] or 'none'} of the kernel's {sites[k][2]} instructions"
                     + (" (the loop over task groups, not a DP loop)" if sites[k][1] > sites[k][2] // 2 else "")) if k in sites else ""
            print(f"  spills: {v:3d} VGPR {s:3d} SGPR, {scratch:5d} B scratch  {names[k][:140]}{where}")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
