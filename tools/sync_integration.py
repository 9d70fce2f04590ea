#!/usr/bin/env python3
"""INTEGRATION.md's listings are literal excerpts of the files under adapters/ and include/ddp/.

A listing is introduced by a marker line
    <!-- excerpt: PATH from="TEXT" to="TEXT" -->
followed by a fenced code block.  The block's body is the lines of PATH from the first line containing the `from` text
through the next line containing the `to` text (both inclusive).  `tools/sync_integration.py` rewrites every block from
the files; `--check` (what tests/test_adapters.py runs) fails when a block and its file have drifted apart."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = re.compile(r'<!-- excerpt: (\S+) from="([^"]*)" to="([^"]*)" -->')


def extract(path, start, stop):
    lines = open(os.path.join(ROOT, path)).read().split("\n")
    a = next(i for i, l in enumerate(lines) if start in l)
    b = next(i for i in range(a, len(lines)) if stop in lines[i])
    return lines[a:b + 1]


def sync(text):
    out, lines, i = [], text.split("\n"), 0
    n_blocks = 0
    while i < len(lines):
        out.append(lines[i])
        m = MARK.match(lines[i].strip())
        if m:
            assert lines[i + 1].startswith("```"), f"marker without a code block: {lines[i]}"
            out.append(lines[i + 1])
            j = i + 2
            while not lines[j].startswith("```"):
                j += 1
            out.extend(extract(m.group(1), m.group(2), m.group(3)))
            out.append(lines[j])
            i = j
            n_blocks += 1
        i += 1
    return "\n".join(out), n_blocks


if __name__ == "__main__":
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    new, n = sync(text)
    if "--check" in sys.argv:
        if new != text:
            sys.exit("INTEGRATION.md listings differ from the files they cite: run tools/sync_integration.py")
        print(f"{n} listings match their files")
    else:
        open(path, "w").write(new)
        print(f"{n} listings written")
