#!/usr/bin/env python3
"""Re-wrap the prose lines of a markdown file to <= WIDTH characters: paragraphs and list items only (tables, code fences, headings and lines that are already short
stay as they are; a wrapped list item continues with the indentation of its text).   python scripts/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence
    if fence or len(line) <= width or line.lstrip().startswith("|") or line.startswith("#"):
        out.append(line)
        continue
    m = re.match(r"^(\s*)((?:[-*+]|\d+[.)]|\([a-z0-9]+\))\s+)?", line)
    lead, bullet = m.group(1), m.group(2) or ""
    body = line[len(lead) + len(bullet):]
    out.extend(textwrap.wrap(body, width=width, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet), break_long_words=False, break_on_hyphens=False))
open(path, "w").write("\n".join(out))
