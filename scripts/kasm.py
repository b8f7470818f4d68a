#!/usr/bin/env python3
"""developer aid: cut one kernel out of a hipcc -save-temps assembly listing.  usage: kasm.py file.s name-substring > out.s"""
import sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
print("\n".join(l for l in lines[start:end + 1] if not l.lstrip().startswith(";")))
