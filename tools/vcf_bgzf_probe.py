"""read_vcf end to end on a bgzip'd VCF-8 file (`x.vcf.gz`: BGZF members of 65 280 bytes, what `bgzip` writes) in the page cache:
COUNT(*), all nine columns (the nested ones made on the device from the inflated text), chrom/pos/ref projected — best of 3 each —
beside the same file as plain text.  VCF_LINES (default 21 M = 1.02 GB of text); EXG_TRACE=2 in the environment for the timeline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from exon_duckdb_amd import device, load_library
    n_lines = int(os.environ.get("VCF_LINES", "21000000"))
    plain, gz = "/dev/shm/exg_bgzf_probe.vcf", "/dev/shm/exg_bgzf_probe.vcf.gz"
    t, n = device.synth_vcf(n_lines)
    bench.write_device_bytes(torch, t, n, plain)
    del t
    torch.cuda.empty_cache()
    from exon_duckdb_amd.testing.bgzf import bgzip
    nz = bgzip(plain, gz)
    lib = load_library()
    for label, p in (("plain text", plain), ("bgzip", gz)):
        bench.reader_count(lib, p, "vcf")
        rows_c, dt_c = min((bench.reader_count(lib, p, "vcf") for _ in range(3)), key=lambda x: x[1])
        rows, chunks, dt_a = min((bench.reader_chunks(lib, p, "vcf") for _ in range(3)), key=lambda x: x[2])
        _, _, dt_p = min((bench.reader_chunks(lib, p, "vcf", columns=0b1011) for _ in range(3)), key=lambda x: x[2])
        assert rows == rows_c == n_lines
        print(f"{label}: {n / 1e9:.2f} GB of VCF-8" + (f" in {nz / 1e9:.2f} GB" if p == gz else "") +
              f": COUNT(*) {dt_c * 1e3:.1f} ms = {n / dt_c / 1e9:.1f} GB/s; all columns {dt_a * 1e3:.1f} ms = {n / dt_a / 1e9:.2f} GB/s; "
              f"chrom,pos,ref {dt_p * 1e3:.1f} ms = {n / dt_p / 1e9:.1f} GB/s", flush=True)
    if os.environ.get("ARROW"):   # the reference's own boundary on both files: new_reader -> Arrow C stream (nested arrays), drained by a C loop
        import ctypes as C
        import time
        from exon_duckdb_amd import load_test_library
        tl = load_test_library()
        for label, p in (("plain text", plain), ("bgzip", gz)):
            best = None
            for _ in range(4):
                rows_a, nb, dg, el = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
                err = C.create_string_buffer(512)
                t0 = time.perf_counter()
                rc = tl.exon_tf_drain_arrow_vcf(p.encode(), None, None, 0, C.byref(rows_a), C.byref(nb), C.byref(dg), C.byref(el), err, 512)
                dt = time.perf_counter() - t0
                assert rc == 0 and rows_a.value == n_lines, err.value
                best = dt if best is None or dt < best else best
            print(f"{label}: new_reader -> Arrow record batches {best * 1e3:.1f} ms = {n / best / 1e9:.2f} GB/s", flush=True)
    os.unlink(plain)
    os.unlink(gz)


if __name__ == "__main__":
    main()
