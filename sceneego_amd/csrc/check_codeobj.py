#!/usr/bin/env python3
"""Build-time guard for the kernels whose LDS-DMAs are issued through inline assembly with HAND-COUNTED `s_waitcnt vmcnt(N)` waits
(conv3d_wino44pp.hip, conv3d_wino67.hip).  Those waits are only right while the compiler puts no vector-memory
operation of its own between the DMAs and the wait, and while nobody but the asm writes M0:

  * every kernel of the object must report .vgpr_spill_count == 0 and .private_segment_fixed_size == 0 (a scratch spill store /
    reload is a VMEM operation the hand-written counts do not know of; SGPR spills go to VGPR lanes - v_writelane / v_readlane -
    and are harmless as long as no scratch exists, which the second key says);
  * in the disassembly M0 may be written ONLY by the `s_mov_b32 m0, sN` that the asm statement itself emits in front of each
    `global_load_lds_dwordx4` (so: as many lines mentioning m0 as LDS-DMA instructions, each directly ahead of its s_nop + DMA).

Usage: check_codeobj.py OBJECT.o [--kernels SUBSTRING]      exit code 0 = clean, 1 = violated (message on stderr).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("SE_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def run(*cmd):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout


def main(argv):
    obj = argv[1]
    want = argv[3] if len(argv) > 3 and argv[2] == "--kernels" else ""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "x.fatbin"), os.path.join(d, "x.co")
        run(f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat)
        run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
            f"--output={co}")
        notes = run(f"{LLVM}/llvm-readelf", "--notes", co)
        asm = run(f"{LLVM}/llvm-objdump", "-d", co)
    bad = []
    # ---- kernel metadata (YAML inside the note): one block per kernel, keys in alphabetical order ----
    kernels = 0
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s*(\S+)", blk).group(1)
        if want and want not in name:
            continue
        kernels += 1
        for key in (".vgpr_spill_count", ".private_segment_fixed_size"):
            m = re.search(re.escape(key) + r":\s*(\d+)", blk)
            if m is None:
                bad.append(f"{name}: metadata key {key} missing")
            elif int(m.group(1)) != 0:
                bad.append(f"{name}: {key} = {m.group(1)} (must be 0: a spill is a vector-memory operation the hand-counted vmcnt waits do not know of)")
    if kernels == 0:
        bad.append(f"no kernel matching '{want}' in {obj}")
    # ---- M0: only the asm statement's own s_mov_b32 in front of a global_load_lds ----
    lines = [l.split("//")[0].strip() for l in asm.splitlines()]
    lines = [l for l in lines if l and not l.endswith(":")]
    n_dma = sum("global_load_lds_dword" in l for l in lines)
    for i, l in enumerate(lines):
        if re.search(r"\bm0\b", l):
            ok = re.match(r"s_mov_b32 m0, s\d+$", l) and i + 2 < len(lines) and lines[i + 1].startswith("s_nop") and \
                "global_load_lds_dword" in lines[i + 2]
            if not ok:
                bad.append(f"M0 touched outside the LDS-DMA asm statement: '{l}'")
    n_m0 = sum(bool(re.search(r"\bm0\b", l)) for l in lines)
    if n_m0 != n_dma:
        bad.append(f"{n_m0} instructions mention m0 but there are {n_dma} global_load_lds_dwordx4")
    if bad:
        print(f"check_codeobj: {obj}:", *bad[:20], sep="\n  ", file=sys.stderr)
        return 1
    print(f"check_codeobj: {os.path.basename(obj)}: {kernels} kernels, no spills, no scratch, {n_dma} LDS-DMAs, M0 written only by them")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
