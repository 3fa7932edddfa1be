#!/usr/bin/env python3
"""Keeps INTEGRATION.md's quoted slot bodies identical to the files they quote: every block between
    <!-- BEGIN integration/<file> -->  and  <!-- END integration/<file> -->
is replaced by the file's text from its first line that is not a comment or an #include on (a ```cpp fence around it).
`python tools/gen_integration_md.py` rewrites INTEGRATION.md; `--check` exits 1 when a block is stale (tests/test_binding.py runs that)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def body_of(path):
    lines = open(os.path.join(ROOT, path)).read().splitlines()
    k = 0
    while k < len(lines) and (lines[k].startswith("//") or lines[k].startswith("#include") or not lines[k].strip()):
        k += 1
    return "\n".join(lines[k:]).rstrip() + "\n"


def render(text):
    def repl(m):
        path = m.group(1)
        return "<!-- BEGIN %s -->\n```cpp\n%s```\n<!-- END %s -->" % (path, body_of(path), path)
    return re.sub(r"<!-- BEGIN (integration/[\w.]+) -->.*?<!-- END \1 -->", repl, text, flags=re.S)


if __name__ == "__main__":
    p = os.path.join(ROOT, "INTEGRATION.md")
    old = open(p).read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    open(p, "w").write(new)
    print("INTEGRATION.md: %d quoted files" % len(re.findall(r"<!-- BEGIN ", new)))
