#!/usr/bin/env python3
"""Build-time guard for the kernels whose LDS-DMAs are issued through inline assembly with HAND-COUNTED `s_waitcnt vmcnt(N)` waits
(conv3d_wino44pp.hip, conv3d_wino67.hip).  Those waits are only right while the compiler puts no vector-memory
operation of its own between the DMAs and the wait, and while nobody but the asm writes M0:

  * every kernel of the object must report .vgpr_spill_count == 0 and .private_segment_fixed_size == 0 (a scratch spill store /
    reload is a VMEM operation the hand-written counts do not know of; SGPR spills go to VGPR lanes - v_writelane / v_readlane -
    and are harmless as long as no scratch exists, which the second key says);
  * in the disassembly M0 may be written ONLY by the `s_mov_b32 m0, sN` that the asm statement itself emits in front of each
    `global_load_lds_dwordx4` (so: as many lines mentioning m0 as LDS-DMA instructions, each directly ahead of its s_nop + DMA).

Usage: check_codeobj.py OBJECT.o [--kernels SUBSTRING]
Exit code 0 = clean, 1 = a condition is VIOLATED, 2 = the check could not run (an LLVM tool is missing, or the code-object metadata did
not parse): build.sh stops on 1 and on 2, with different messages; SE_SKIP_CODEOBJ_CHECK=1 skips the check (exit 0, says so).
The tools are looked up in $SE_LLVM_BIN, then `hipconfig --rocmpath`/lib/llvm/bin, /opt/rocm/lib/llvm/bin, then PATH.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile


class CannotCheck(Exception):
    pass


def tool(name):
    dirs = [os.environ.get("SE_LLVM_BIN")]
    try:
        dirs.append(os.path.join(subprocess.run(["hipconfig", "--rocmpath"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                                timeout=20).stdout.strip(), "lib", "llvm", "bin"))
    except (OSError, subprocess.SubprocessError):
        pass
    dirs.append("/opt/rocm/lib/llvm/bin")
    for d in dirs:
        if d and os.path.isfile(os.path.join(d, name)):
            return os.path.join(d, name)
    p = shutil.which(name)
    if p is None:
        raise CannotCheck(f"{name} not found (set SE_LLVM_BIN to the directory of the ROCm LLVM tools)")
    return p


def run(*cmd):
    try:
        return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout
    except subprocess.CalledProcessError as e:
        raise CannotCheck(f"{os.path.basename(cmd[0])} failed: {e.stderr.strip()[-300:]}")


def main(argv):
    obj = argv[1]
    want = argv[3] if len(argv) > 3 and argv[2] == "--kernels" else ""
    if os.environ.get("SE_SKIP_CODEOBJ_CHECK") == "1":
        print(f"check_codeobj: {os.path.basename(obj)}: SKIPPED (SE_SKIP_CODEOBJ_CHECK=1)")
        return 0
    try:
        return check(obj, want)
    except CannotCheck as e:
        print(f"check_codeobj: {obj}: the check could NOT RUN - {e}", file=sys.stderr)
        return 2


def check(obj, want):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "x.fatbin"), os.path.join(d, "x.co")
        run(tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat)
        run(tool("clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
            f"--output={co}")
        notes = run(tool("llvm-readelf"), "--notes", co)
        asm = run(tool("llvm-objdump"), "-d", co)
    bad = []
    # ---- kernel metadata (YAML inside the note): one block per kernel; split at the list items that carry a `.name` ----
    kernels = 0
    blocks = [b for b in re.split(r"\n\s*- (?=\.[a-z_]+:)", notes) if re.search(r"\.name:\s*\S+", b) and ".vgpr_count" in b]
    if not blocks:
        raise CannotCheck("no kernel metadata block found in the code object's notes (format changed?)")
    for blk in blocks:
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
