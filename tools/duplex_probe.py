"""Is the host link full duplex for this process?  exon_tf_link_probe (csrc/testing/exg_synth.hip): pinned 1 GiB buffers each
way — H2D alone, D2H alone, both at once, D2H by a copy kernel (stores into the pinned block) alone and beside an SDMA H2D,
two D2H at once.  What the compressed-input path into DataChunks does all the time: compressed windows up, decoded segments +
column vectors down."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import load_test_library
tl = load_test_library()
for blocks in (int(x) for x in os.environ.get("COPY_BLOCKS", "64,256,1024").split(",")):
    out = (C.c_double * 6)()
    assert tl.exon_tf_link_probe(0, 1 << 30, blocks, out) == 0
    g = [v / 1e9 for v in out]
    print(f"copy kernel of {blocks} blocks: H2D {g[0]:.1f}  D2H {g[1]:.1f}  H2D+D2H {g[2]:.1f} (aggregate)  D2H-kernel {g[3]:.1f}  "
          f"H2D+D2H-kernel {g[4]:.1f} (aggregate)  D2H+D2H {g[5]:.1f} GB/s", flush=True)
