#!/usr/bin/env python3
"""Where the packed kernels' scratch instructions are: the general path each of them carries as its TAIL (tail_instance, inlined: DESIGN.md §3.11)
is told apart from the packed path by the line tables of the device-only listing (hipcc -S -g1): see analyse(). Prints, per kernel variant,
the scratch_load / scratch_store instructions on the packed path and in the tail and, with -v, the source lines the packed path's belong to.
CPU only (cross-compiles):
    python tools/hot_path_spills.py [-v] [family.part ...]      default: sim3p.0 orthp.0 orthp.1 boxp.0   (WBC_XFLAGS="-D..." adds compiler flags)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd", "csrc")


def listing(part):
    fam, k = part.split(".")
    out = os.path.join(tempfile.gettempdir(), "wbc_%s_%s.s" % (fam, k))
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S", "-g1",
                           "-D%s_PART=%s" % (fam.upper(), k)] + os.environ.get("WBC_XFLAGS", "").split() + [os.path.join(CSRC, "wbc_k_%s.hip" % fam), "-o", out])
    return open(out).read()


def analyse(part, verbose):
    """-> (part, [(kernel, [packed path loads, stores], [tail loads, stores], {source line: count on the packed path})])
    Control-flow split of the listing: basic blocks from the labels and s_branch / s_cbranch targets; the packed path is what is reachable
    from the kernel's entry WITHOUT passing the `; WBC_TAIL_BEGIN` comment tail_instance opens with (layout order does not help: the compiler
    moves blocks; line tables do not either: helpers such as cross3 are inlined on both sides)."""
    txt = listing(part)
    files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt))
    res = []
    for m in re.finditer(r"^(_ZN3wbc\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
        name = re.sub(r"\(.*", "", name.replace("void wbc::", ""))
        # blocks: list of dicts(label, insts [(text, loc)], succ labels, falls through, has_marker_at index)
        blocks, cur, loc = [], {"label": "entry", "ins": [], "succ": [], "fall": True, "cut": None}, None
        for line in m.group(2).splitlines():
            lm = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
            if lm:
                loc = (os.path.basename(files.get(lm.group(1), "?")), int(lm.group(2)))
                continue
            lab = re.match(r"^(\.LBB\w+):", line)
            if lab:
                blocks.append(cur)
                cur = {"label": lab.group(1), "ins": [], "succ": [], "fall": True, "cut": None}
                continue
            if "WBC_TAIL_BEGIN" in line and cur["cut"] is None:
                cur["cut"] = len(cur["ins"])
                continue
            ins = line.split(";")[0].strip()
            if not ins or ins.startswith("."):
                continue
            cur["ins"].append((ins, loc))
            br = re.match(r"s_(c?branch\w*)\s+(\.LBB\w+)", ins)
            if br:
                cur["succ"].append(br.group(2))
                if br.group(1) == "branch":
                    cur["fall"] = False
            if ins.startswith("s_endpgm") or ins.startswith("s_setpc"):
                cur["fall"] = False
        blocks.append(cur)
        index = {b["label"]: i for i, b in enumerate(blocks)}
        # an unconditional s_branch may be followed by dead text in the same block: instructions after it still belong to the block (harmless)
        seen, stack = set(), [0]
        while stack:
            i = stack.pop()
            if i in seen or i >= len(blocks):
                continue
            seen.add(i)
            b = blocks[i]
            if b["cut"] is not None:
                continue                     # everything behind the marker is the tail
            for t in b["succ"]:
                stack.append(index[t])
            if b["fall"]:
                stack.append(i + 1)
        hot, tail, where = [0, 0], [0, 0], {}
        for i, b in enumerate(blocks):
            for k_, (ins, lc) in enumerate(b["ins"]):
                k = 0 if ins.startswith("scratch_load") else (1 if ins.startswith("scratch_store") else -1)
                if k < 0:
                    continue
                is_hot = i in seen and (b["cut"] is None or k_ < b["cut"])
                (hot if is_hot else tail)[k] += 1
                if is_hot:
                    where[lc] = where.get(lc, 0) + 1
        res.append((name, hot, tail, where))
    return part, res


def main():
    parts = [a for a in sys.argv[1:] if not a.startswith("-")] or ["sim3p.0", "orthp.0", "orthp.1", "boxp.0"]
    verbose = "-v" in sys.argv
    with ThreadPoolExecutor(max_workers=min(4, len(parts))) as ex:
        for part, res in ex.map(lambda p: analyse(p, verbose), parts):
            for name, hot, tail, where in res:
                print("%-46s packed path: %3d scratch_load %3d scratch_store | tail (general path): %3d / %3d" % (name, hot[0], hot[1], tail[0], tail[1]))
                if verbose:
                    for (f, ln), n in sorted(where.items(), key=lambda t: (t[0] is None, t[0])):
                        print("      %s:%s  x%d" % (f, ln, n))


if __name__ == "__main__":
    main()
