"""sha256/16 over the library's sources (csrc/*.h, csrc/*.hip, include/*.h; file name + contents, ordered by file name).

csrc/build.sh bakes the value into the binary (`lmi_build_info()`), bench.py reports it for the library it LOADED, and
profiles/summarize.py stamps PMC summaries with the value the profiled bench run reported -- so a summary is replayed into a
bench line only for the very build it was collected from.  Run as a script it prints the hash of the working tree."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_sha16(root: str = os.path.dirname(_HERE)) -> str:
    files = (glob.glob(os.path.join(root, "learnedmetricindex_amd", "csrc", "*.h"))
             + glob.glob(os.path.join(root, "learnedmetricindex_amd", "csrc", "*.hip"))
             + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in sorted(files, key=os.path.basename):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def file_sha16(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
