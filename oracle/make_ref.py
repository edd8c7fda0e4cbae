"""TEST INFRASTRUCTURE ONLY (oracle/): make the *unmodified* Python reference travel to the GPU box.

/root/reference exists in the build container only.  This recipe packs the reference's own package directory
(naturalspeech2_pytorch/*.py and utils/, source untouched) into oracle/_ref/reference_py.tar.gz.  oracle/_ref/ is listed in
.gitignore (nothing of the reference enters the repository history) but not in .gpurunignore, so the archive travels with the
snapshot like the built .so files.  oracle/ref_stub.py unpacks it into a temporary directory when /root/reference is absent,
so on the MI355X box
  * tests/test_reference_gpu.py runs the reference's own NaturalSpeech2 (sampler, loss) around compat.HipBackedModel, and
  * bench.py's cpu_baseline times the reference's own Model (kind "reference") instead of the oracle port.
Run by __graft_entry__.build() whenever /root/reference is present:   python oracle/make_ref.py
"""
import os
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("NS2_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "reference_py.tar.gz")
PKG = "naturalspeech2_pytorch"


def make() -> str:
    pkg = os.path.join(SRC, PKG)
    if not os.path.isdir(pkg):
        raise RuntimeError(f"reference not present at {SRC}")
    os.makedirs(OUT_DIR, exist_ok=True)

    def keep(ti):
        name = os.path.basename(ti.name)
        if "__pycache__" in ti.name or name.endswith((".pyc", ".pyo")):
            return None
        ti.mtime = 0            # reproducible archive
        ti.uid = ti.gid = 0
        ti.uname = ti.gname = ""
        return ti

    tmp = ARCHIVE + ".tmp"
    with tarfile.open(tmp, "w:gz") as tf:
        tf.add(pkg, arcname=PKG, filter=keep)
    os.replace(tmp, ARCHIVE)
    return ARCHIVE


if __name__ == "__main__":
    print("wrote", make(), os.path.getsize(ARCHIVE), "bytes")
    sys.exit(0)
