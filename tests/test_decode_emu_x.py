"""The experiment switches of kafka_decode_coop (csrc/kta_decode_coop.h: DX_*) under the fibre emulator: window bases on 128-byte lines,
the two prefetching forms — against the encoder, the oracle and (records that both deliver) the host statement of the plain rounds."""
import numpy as np
import pytest

from kafka_cases import assert_columns
import test_decode_rounds as R
from test_decode_emu import ORDERS, _awkward_blobs, _random_case, emu, run_kernel  # noqa: F401

XS = [2, 64, 66, 128, 130, 256, 258, 512, 514, 770, 642, 1024, 1282]
GEOS = [(16, 3072, 16), (32, 8192, 32), (8, 256, 8), (8, 1024, 16)]


@pytest.mark.parametrize("x", XS)
@pytest.mark.parametrize("geometry", GEOS)
@pytest.mark.parametrize("seed,with_keys,max_records", [(1, True, 40), (2, False, 40), (3, True, 700), (5, True, 300), (6, True, 3000)])
def test_switches_match_encoder_and_oracle(emu, seed, with_keys, max_records, geometry, x):
    if x & 256 and geometry[1] % 1024:
        pytest.skip("1 KiB loads want whole-KiB windows")
    if x & (512 | 1024) and geometry[2] != geometry[0]:
        pytest.skip("staged columns: one record per lane and round")
    blob, expected, want = _random_case(seed, max_records)
    order = ORDERS[(seed + geometry[1] // 64 + x) % len(ORDERS)]
    cols, bad = run_kernel(emu, blob, 3, geometry, order, with_keys=with_keys, with_seq=(seed % 2 == 1), seq_base=10**12, prefetch=x)
    assert bad == 0
    assert_columns(cols, expected)
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k
    assert cols["n_key_bytes"] == int(np.maximum(want["key_len"], 0).sum())


@pytest.mark.parametrize("x", XS)
@pytest.mark.parametrize("geometry", GEOS)
def test_switches_report_what_the_plain_rounds_report(emu, geometry, x):
    if x & 256 and geometry[1] % 1024:
        pytest.skip("1 KiB loads want whole-KiB windows")
    if x & (512 | 1024) and geometry[2] != geometry[0]:
        pytest.skip("staged columns: one record per lane and round")
    for n, blob in enumerate(_awkward_blobs()):
        want, _, _, want_bad = R.rounds_host(blob, 1, geometry)
        for order, poison in ((ORDERS[n % 3], 0xEE), (ORDERS[(n + 1) % 3], 0x00 if n % 2 else 0xFF)):
            cols, bad = run_kernel(emu, blob, 1, geometry, order, poison, prefetch=x)
            assert bad == want_bad, (n, order)
            both = (cols["partition"] != -1) & (want["partition"] != -1)
            if x in (512, 1024):                          # the plain rounds, only the stores differ: the same records delivered
                both = np.ones(len(both), bool)
            for k in ("partition", "key_len", "val_len", "ts_ms", "key_off"):
                assert np.array_equal(cols[k][both], want[k][both]), (n, order, k)
