"""CPU-only: the index file format (annhip_save_write/read).  Host code, no GPU needed."""
import os

import numpy as np
import pytest

import approximatenn_amd as A
from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, load_golden


@pytest.mark.parametrize("name", ["defaults_d80_f32", "odd_everything_f64", "tiny_appendixA_f32", "k17_d100_f64"])
def test_roundtrip_and_cross_consumption(name, tmp_path):
    g = load_golden(name)
    save = A.Save.from_dict(g["prec"], g["save"])
    path = tmp_path / "index.ann"
    save.write(path)
    back = A.Save.read(g["prec"], path)
    try:
        loaded = back.to_dict()
        assert_save_equal(loaded, g["save"])
        # a loaded index is consumable by the CPU path: same answers as the reference recorded
        orc = O.CpuBackend(g["prec"], "oracle")
        ids, dd = orc.query(loaded, g["points"], g["y"])
        assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"])
    finally:
        back.free()
    # ids are stored as 32 bit: the file is markedly smaller than the in-memory size_t arrays
    mem = sum(w.size for w in g["save"]["which_par"]) * 8 + g["save"]["graph"].size * 8
    assert os.path.getsize(path) < 0.75 * mem + 4096 + g["save"]["bases"].nbytes


def test_rejects_wrong_precision_truncation_and_corruption(tmp_path):
    g = load_golden("tiny_appendixA_f32")
    save = A.Save.from_dict("f32", g["save"])
    path = tmp_path / "index.ann"
    save.write(path)
    with pytest.raises(OSError):
        A.Save.read("f64", path)                      # written by the float build
    blob = path.read_bytes()
    (tmp_path / "short.ann").write_bytes(blob[: len(blob) // 2])
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "short.ann")
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x40
    (tmp_path / "bad.ann").write_bytes(bytes(bad))
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "bad.ann")
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "missing.ann")
    (tmp_path / "junk.ann").write_bytes(b"not an index file at all" * 10)
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "junk.ann")


def test_header_is_checked_against_the_file_size_before_anything_is_allocated(tmp_path):
    """A header that promises more than the file holds (n, par_maxes, d_long blown up) must be refused up front --
    no multi-terabyte malloc, no fread into NULL -- and so must a file with bytes appended."""
    import struct
    g = load_golden("tiny_appendixA_f32")
    save = A.Save.from_dict("f32", g["save"])
    path = tmp_path / "index.ann"
    save.write(path)
    blob = path.read_bytes()
    assert blob[:8] == b"ANNSAVE2"
    T = int(g["save"]["tries"])
    for off, val in ((24, 1 << 40),            # n
                     (48, 1 << 23),            # d_long
                     (56, 1 << 50),            # par_maxes[0]
                     (32, 1 << 30),            # k (then n <= k)
                     (16, 4000)):              # tries
        bad = bytearray(blob)
        bad[off:off + 8] = struct.pack("<Q", val)
        (tmp_path / "hdr.ann").write_bytes(bytes(bad))
        with pytest.raises(OSError):
            A.Save.read("f32", tmp_path / "hdr.ann")
    (tmp_path / "long.ann").write_bytes(blob + b"\0" * 16)
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "long.ann")
    assert T >= 1


def test_ids_outside_their_range_are_refused_even_with_a_valid_checksum(tmp_path):
    g = load_golden("tiny_appendixA_f32")
    arrays = dict(g["save"])
    arrays["which_par"] = [w.copy() for w in arrays["which_par"]]
    arrays["which_par"][0][0, 0] = int(arrays["n"]) + 7          # neither a point id nor the padding n
    save = A.Save.from_dict("f32", arrays)
    path = tmp_path / "index.ann"
    save.write(path)                                             # the writer does not judge; the checksum is valid
    with pytest.raises(OSError):
        A.Save.read("f32", path)


def test_checksum_throughput(tmp_path):
    """The word-wise checksum must not be the bottleneck of multi-GB index files (format 1's byte-wise FNV was)."""
    import time
    n, k, ds, T = 200_000, 10, 14, 2
    rng = np.random.default_rng(0)
    arrays = dict(tries=T, n=n, k=k, d_short=ds, d_long=16, par_maxes=np.array([40, 40], dtype=np.uint64),
                  graph=rng.integers(0, n, size=(n, k), dtype=np.uint64),
                  which_par=[rng.integers(0, n + 1, size=(1 << ds, 40), dtype=np.uint64) for _ in range(T)],
                  row_means=np.zeros(16, dtype=np.float32), bases=np.zeros((T, ds, 16), dtype=np.float32))
    save = A.Save.from_dict("f32", arrays)
    path = tmp_path / "big.ann"
    t0 = time.perf_counter()
    save.write(path)
    t1 = time.perf_counter()
    back = A.Save.read("f32", path)
    t2 = time.perf_counter()
    try:
        assert_save_equal(back.to_dict(), arrays)
    finally:
        back.free()
    mb = os.path.getsize(path) / 1e6
    assert mb / (t1 - t0) > 100 and mb / (t2 - t1) > 100, (mb, t1 - t0, t2 - t1)   # MB/s, generous lower bound


def test_format_1_files_are_still_read(tmp_path):
    """Index files of the library's first format (magic ANNSAVE1, FNV-1a-64 checksum over the same layout) load like
    format 2 -- behind the same size, checksum and id-range checks."""
    g = load_golden("tiny_appendixA_f32")
    save = A.Save.from_dict("f32", g["save"])
    path = tmp_path / "index2.ann"
    save.write(path)
    body = bytearray(open(path, "rb").read()[:-8])
    body[:8] = b"ANNSAVE1"
    h = 1469598103934665603
    for b in body:                                   # FNV-1a, 64 bit
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    old = tmp_path / "index1.ann"
    old.write_bytes(bytes(body) + h.to_bytes(8, "little"))
    back = A.Save.read("f32", old)
    try:
        assert_save_equal(back.to_dict(), g["save"])
    finally:
        back.free()
    bad = bytearray(old.read_bytes())
    bad[200] ^= 1                                    # a flipped bit: the old checksum catches it too
    (tmp_path / "bad1.ann").write_bytes(bytes(bad))
    with pytest.raises(OSError):
        A.Save.read("f32", tmp_path / "bad1.ann")
