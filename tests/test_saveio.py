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
