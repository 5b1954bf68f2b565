#!/usr/bin/env python3
"""Round 6: remove the compile-time A/B and ablation branches whose verdicts are recorded (profiles/r0*_experiments.md) from lpi_amd/csrc — a minimal `unifdef`
for the macros listed below, all treated as UNDEFINED (the product build never defined them), so the compiled device code is unchanged (checked: the
.hip_fatbin sections of every object are byte-identical before and after).  The ablation builds themselves are history: the last tree that has them is the
parent of the commit that introduced this script.  Usage: python tools/probe/strip_settled_macros.py [--check]"""
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "lpi_amd", "csrc")
UNDEF = {
    "LPI_ABL_ATTN_NOLOAD", "LPI_ABL_ATTN_NOCOMPUTE", "LPI_ABL_SHARED_NOPARTIAL", "LPI_ABL4_NOCOMPUTE", "LPI_ABL4_NODMA", "LPI_ABL4_NOSTORE", "LPI_ABL4_NODQST",
    "LPI_ABL4_NODKVST", "LPI_NT_ATTN", "LPI_NO_NT_SIDE", "LPI_ABL_NO_STAGING_WRITE", "LPI_NT_SIDE16C", "LPI_EPI_SLEEP", "LPI_RS_NOREDUCE", "LPI_RS_NOSTORE",
    "LPI_GROUP_PAD8", "LPI_NT_LN_LD", "LPI_NT_LN", "LPI_ABL_ATTN_NOSUM", "LPI_GELU_GRAD_V1", "LPI_ABL_NO_GLOBAL_STORE", "LPI_LN_PLAIN_C", "LPI_SCALAR_GELU",
}
DIR = re.compile(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$")


def strip(text):
    out, stack = [], []      # stack entries: (kind, keep_now, ours, seen_else)  kind: 'ours' | 'other'
    for line in text.split("\n"):
        m = DIR.match(line)
        emitting = all(k for _, k, _, _ in stack)
        if m:
            d, rest = m.group(1), m.group(2)
            name = rest.strip().split()[0] if rest.strip() else ""
            name = re.sub(r"/\*.*", "", name).strip()
            if d in ("ifdef", "ifndef") and name in UNDEF:
                stack.append(("ours", (d == "ifndef"), True, False))
                continue
            if d == "if" and re.fullmatch(r"\s*defined\((\w+)\)\s*(/\*.*\*/)?\s*", rest) and re.search(r"defined\((\w+)\)", rest).group(1) in UNDEF:
                stack.append(("ours", False, True, False))
                continue
            if d in ("ifdef", "ifndef", "if"):
                stack.append(("other", True, False, False))
                if emitting:
                    out.append(line)
                continue
            if d in ("else", "elif"):
                kind, keep, ours, seen = stack[-1]
                if ours:
                    if d == "elif":
                        # '#elif defined(OTHER)' behind one of ours: becomes '#if' of a foreign block when ours was false
                        mm = re.search(r"defined\((\w+)\)", rest)
                        if mm and mm.group(1) in UNDEF:
                            stack[-1] = (kind, False, True, True)
                        elif not keep:
                            stack[-1] = ("other", True, False, False)
                            if all(k for _, k, _, _ in stack[:-1]):
                                out.append(re.sub(r"#\s*elif", "#if", line, 1))
                        else:
                            raise SystemExit("elif after a kept branch of a settled macro: handle by hand: " + line)
                    else:
                        stack[-1] = (kind, not keep, True, True)
                    continue
                if all(k for _, k, _, _ in stack[:-1]):
                    out.append(line)
                continue
            if d == "endif":
                kind, keep, ours, seen = stack.pop()
                if not ours and all(k for _, k, _, _ in stack):
                    out.append(line)
                continue
        if emitting:
            out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    check = "--check" in sys.argv
    for fn in sorted(os.listdir(ROOT)):
        if not fn.endswith((".hip", ".h")):
            continue
        p = os.path.join(ROOT, fn)
        src = open(p).read()
        new = strip(src)
        if new != src:
            print(f"{fn}: {src.count(chr(10)) - new.count(chr(10))} lines removed")
            if not check:
                open(p, "w").write(new)
