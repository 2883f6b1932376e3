#!/usr/bin/env python3
"""Reflow the prose of a markdown file to <= WIDTH characters per line: consecutive prose lines of a paragraph or list item are joined and wrapped again (a wrapped
list item continues with the indentation of its text).  Tables, code fences, headings and blank lines stay as they are.   python scripts/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
BULLET = re.compile(r"^(\s*)((?:[-*+]|\d+[.)]|\d+[a-z][.)])\s+)")
out, fence, group = [], False, None   # group = [lead, bullet, [text parts]]


def flush():
    global group
    if group:
        lead, bullet, parts = group
        body = " ".join(p.strip() for p in parts)
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet), break_long_words=False, break_on_hyphens=False) or [lead + bullet])
    group = None


for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        flush(); fence = not fence; out.append(line); continue
    if fence or not line.strip() or line.lstrip().startswith("|") or line.startswith("#"):
        flush(); out.append(line); continue
    m = BULLET.match(line)
    if m:
        flush(); group = [m.group(1), m.group(2), [line[m.end():]]]
    elif group is not None:
        group[2].append(line)
    else:
        lead = re.match(r"^\s*", line).group(0)
        group = [lead, "", [line]]
flush()
open(path, "w").write("\n".join(out))
