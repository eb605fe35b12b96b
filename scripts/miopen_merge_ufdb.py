"""MIOpen keeps one user find-db per library VERSION (<arch>.HIP.<version>.ufdb.txt).  This image has two MIOpen builds -- the one
inside the torch wheel (3.5.0-...) and /opt/rocm's (3.5.1, what MIOpenDriver links) -- so find records made by MIOpenDriver are
invisible to a PyTorch process.  The record format is the same: this writes the union of all *.ufdb.txt files of a directory
(by problem key; a later file's record replaces an earlier one's) into each of them.  python scripts/miopen_merge_ufdb.py <db dir>"""
import glob, os, sys
d = sys.argv[1]
files = sorted(glob.glob(os.path.join(d, "*.ufdb.txt")), key=os.path.getmtime)
rec = {}
for fn in files:
    for line in open(fn):
        line = line.rstrip("\n")
        if "=" in line:
            rec[line.split("=", 1)[0]] = line
for fn in files:
    with open(fn, "w") as f:
        f.write("\n".join(rec[k] for k in sorted(rec)) + "\n")
print(f"{len(rec)} find records in each of {len(files)} files: {[os.path.basename(f) for f in files]}")
