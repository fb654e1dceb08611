"""A single-member gzip of more than 4 GiB COMPRESSED through the reader (bit positions inside a decode are rebased,
so chunks that start beyond 4 GiB of input must decode like the others).  The member is written pigz-style: parts
deflated in parallel, each closed with a sync flush, concatenated into one DEFLATE stream."""
import multiprocessing as mp, os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PART = 64 << 20
RAW = "/tmp/exg_big.fastq"
GZ = "/tmp/exg_big.fastq.gz"


def deflate_part(args):
    off, n, last = args
    with open(RAW, "rb") as f:
        f.seek(off)
        data = f.read(n)
    co = zlib.compressobj(int(os.environ.get("GZ_LEVEL", "6")), zlib.DEFLATED, -15)
    return co.compress(data) + (co.flush(zlib.Z_FINISH) if last else co.flush(zlib.Z_SYNC_FLUSH))


if __name__ == "__main__":
    from exon_duckdb_amd import device, table_function
    n_rec = int(float(os.environ.get("GZ_BIG_GB", "9.2")) * 1e9) // 332
    nb = 332 * n_rec
    crc = 0
    with open(RAW, "wb") as f:
        step = 332 * 3_000_000
        for off in range(0, nb, step):
            n = min(step, nb - off)
            part = device.synth_fastq(n, file_offset=off)[:n].cpu().numpy().tobytes()
            crc = zlib.crc32(part, crc)
            f.write(part)
    parts = [(o, min(PART, nb - o), o + PART >= nb) for o in range(0, nb, PART)]
    t0 = time.time()
    with mp.get_context("fork").Pool(min(64, os.cpu_count())) as pool, open(GZ, "wb") as out:
        out.write(b"\x1f\x8b\x08\x00" + b"\0" * 4 + b"\0\xff")
        for blob in pool.imap(deflate_part, parts):
            out.write(blob)
        out.write(struct.pack("<II", crc, nb & 0xFFFFFFFF))  # both are verified by the reader (CRC-32 of 9 GB folded from 64 KiB segments)
    comp = os.path.getsize(GZ)
    print(f"{nb / 1e9:.2f} GB of FASTQ -> {comp / 1e9:.2f} GB single-member gzip in {time.time() - t0:.0f} s", flush=True)
    os.unlink(RAW)
    con = table_function.connect()
    rel = con.table_function("read_fastq", GZ)
    for _ in range(2):
        t0 = time.time(); n = rel.count(); dt = time.time() - t0
        assert n == n_rec, (n, n_rec)
        print(f"COUNT(*): {dt:.3f} s = {nb / dt / 1e9:.1f} GB/s of FASTQ", flush=True)
    os.unlink(GZ)
