"""quality_score_string_to_list on device-resident columns (C-ABI exg_quality_score_list) against the oracle.

Bit-exact bar: the list_entry_t {offset, length} of every row and the child INTEGER vector equal what the oracle's
restatement of fastq_functions/module.cpp:28-54 gives for the same strings.
"""
import numpy as np
import pytest

from exon_duckdb_amd import abi

pytestmark = pytest.mark.gpu

BASE = 0x7F0000000000


def scan_fastq(data, capacity=None):
    from exon_duckdb_amd import device

    d_in = device.upload(bytes(data))
    scan = device.FastqScan(len(data), capacity_records=capacity)
    scan.launch(d_in, payload_base=BASE)
    res = scan.fetch()
    assert res.error_code == 0
    return d_in, scan, int(res.n_records)


def check_column(oracle, strings, n, d_payload, col):
    from exon_duckdb_amd import device

    entries, values, total = device.quality_score_string_to_list(strings, n, d_payload, BASE)
    want_entries, want_values = oracle.quality_score_string_to_list(col)
    assert total == len(want_values)
    assert np.array_equal(entries.cpu().numpy().view(np.uint64), want_entries)
    assert np.array_equal(values.cpu().numpy(), want_values)


def test_golden_fastq_quality_list(gpu, oracle, golden_dir):
    data = open(f"{golden_dir}/test.fastq", "rb").read()
    d_in, scan, n = scan_fastq(data)
    exp = oracle.fastq_parse(data, payload_base=BASE)
    assert n == exp.n_rows == 2
    check_column(oracle, scan.cols[3], n, d_in, exp.columns["quality_scores"])
    from exon_duckdb_amd import device

    entries, values, total = device.quality_score_string_to_list(scan.cols[3], n, d_in, BASE)
    q0 = exp.columns["quality_scores"].row(0)
    assert values.cpu().numpy()[: len(q0)].tolist() == [c - 33 for c in q0]      # '!' -> 0, 'I' -> 40
    assert entries.cpu().numpy()[0].tolist() == [0, len(q0)]


@pytest.mark.parametrize("n_rec", [1, 255, 256, 257, 100_000])
def test_synthetic_fastq_all_columns(gpu, oracle, n_rec):
    data = oracle.synth_fastq(332 * n_rec)
    d_in, scan, n = scan_fastq(data)
    exp = oracle.fastq_parse(data, payload_base=BASE)
    assert n == n_rec
    # quality (150 B, pointer form), description (10 B, inlined form), name (15 B)
    for k, name in ((3, "quality_scores"), (1, "description"), (0, "name")):
        check_column(oracle, scan.cols[k], n, d_in, exp.columns[name])


def test_ragged_with_null_rows(gpu, oracle):
    data = oracle.synth_fastq_ragged(20_000)
    d_in, scan, n = scan_fastq(data)
    exp = oracle.fastq_parse(data, payload_base=BASE)
    assert n == exp.n_rows
    d = exp.columns["description"]
    assert (d.valid == 0).any()
    check_column(oracle, scan.cols[1], n, d_in, d)      # NULL rows -> empty entries
    check_column(oracle, scan.cols[3], n, d_in, exp.columns["quality_scores"])


class _Col:
    def __init__(self, offsets, values):
        self.offsets, self.values, self.valid = offsets, values, None


def hand_made_column(lengths, seed):
    """string_t rows (DuckDB v0.8.1 layout) over random bytes 0..255, built on the host."""
    rng = np.random.default_rng(seed)
    lengths = np.asarray(lengths, np.int64)
    offsets = np.zeros(len(lengths) + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    payload = rng.integers(0, 256, int(offsets[-1]) + 16, dtype=np.uint8)
    st = np.zeros((len(lengths), 16), np.uint8)
    for r, (o, ln) in enumerate(zip(offsets[:-1], lengths)):
        st[r, :4] = np.frombuffer(np.uint32(ln).tobytes(), np.uint8)
        if ln <= 12:
            st[r, 4:4 + ln] = payload[o:o + ln]
        else:
            st[r, 4:8] = payload[o:o + 4]
            st[r, 8:16] = np.frombuffer(np.uint64(BASE + o).tobytes(), np.uint8)
    return st, payload, _Col(offsets, payload[: int(offsets[-1])])


@pytest.mark.parametrize("seed,lengths", [
    (1, [0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 0, 0, 7]),
    (2, list(np.random.default_rng(7).integers(0, 40, 5000))),
    (3, [9000, 3, 0, 70_000, 12, 13] + [1] * 600),          # strings longer than a workgroup's stride
    (4, [0] * 1000),
])
def test_hand_made_columns_signed_bytes(gpu, oracle, seed, lengths):
    import torch

    from exon_duckdb_amd import device

    st, payload, col = hand_made_column(lengths, seed)
    d_payload = device.upload(payload.tobytes())
    strings = torch.from_numpy(st.view(np.int64).reshape(-1, 2).copy()).cuda()
    check_column(oracle, strings, len(lengths), d_payload, col)
    if len(col.values):
        _, want = oracle.quality_score_string_to_list(col)
        assert want.min() >= -128 - 33 and want.max() <= 127 - 33
        if (col.values >= 0x80).any():
            assert want.min() < -33          # bytes >= 0x80 are negative chars


def test_capacity_too_small_writes_nothing(gpu, oracle):
    import torch

    from exon_duckdb_amd import device

    st, payload, col = hand_made_column([20] * 300, 5)
    d_payload = device.upload(payload.tobytes())
    strings = torch.from_numpy(st.view(np.int64).reshape(-1, 2).copy()).cuda()
    entries, values, total = device.quality_score_string_to_list(strings, 300, d_payload, BASE, values_capacity=5999)
    assert total == 6000 and len(values) == 5999
    entries, values, total = device.quality_score_string_to_list(strings, 300, d_payload, BASE, values_capacity=6000)
    assert total == 6000
    assert np.array_equal(values.cpu().numpy(), oracle.quality_score_string_to_list(col)[1])


def test_empty_column_and_bad_arguments(gpu):
    import ctypes as C

    import torch

    from exon_duckdb_amd import device
    from exon_duckdb_amd._lib import ExgError

    strings = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
    entries, values, total = device.quality_score_string_to_list(strings, 0, strings, BASE, values_capacity=16)
    assert total == 0 and len(entries) == 0 and len(values) == 0
    a = abi.QualityListArgs()
    with pytest.raises(ExgError):
        device.check(gpu.exg_quality_score_list(C.byref(a)))
