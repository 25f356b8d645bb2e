"""The one reference-PRODUCED vector this repo holds: /root/reference/demo_output.png (README.md:28), a real run
of the reference against a 10-partition topic, transcribed into tests/golden/reference_demo_output.json
(tests/golden/make_reference_demo.py carries the transcription and its self-checks).

What it pins (rows a8 and f1 of SURVEY §8): floor division of the three averages by `alive`
(metric.rs:132-157: 6553301273 / 25056009 -> 261, partition 8 -> 262 / 271), `0.0000` for the dirty ratio of
a partition without tombstones (metric.rs:159-167, main.rs:161), `Estimated Msg/s` = overall_count / secs and
`Topic Size` = overall_size (main.rs:130,137), `P-Bytes` (main.rs:165), the DateTime<Utc> display
(main.rs:132-133), the 120-character rules and the prettytable geometry (main.rs:126-178).

What it cannot pin: handle_message itself — the topic behind the screenshot is gone.  The record stream used
below is OURS: a stream constructed so that the reference's counters come out as the screenshot's input
columns; the oracle / the GPU path must then reproduce the counters from it and every derived value from
the counters.

The screenshot's build printed `|< OS` / `>| OS`; src/main.rs:150,175 of the v0.5.0 source print `< OS` /
`> OS`.  The product prints the v0.5.0 names; the comparison with the screenshot substitutes the two names
(and nothing else: the column is one character wider in the screenshot for that reason alone)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from kafka_topic_analyzer_amd import _native as N
from helpers import NOW, ROOT, load_golden
from oracle_c import Oracle

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402

G = load_golden("reference_demo_output.json")
P = len(G["partitions"])
SCREEN_OS = ("|< OS", ">| OS")


def to_v050(lines):
    """The screenshot's text with the two column names of src/main.rs:150,175 (v0.5.0)."""
    out = []
    for l in lines:
        if l.startswith("+---+"):
            l = l.replace("+---+-------+", "+---+------+", 1)
        elif l.startswith("| P |"):
            l = l.replace("| |< OS | >| OS     |", "| < OS | > OS      |", 1)
        elif l.startswith("| DR"):
            l = l.replace("|< OS", "< OS").replace(">| OS", "> OS")
        elif l.startswith("| ") and l[2].isdigit():
            l = l.replace("| 0     |", "| 0    |", 1)
        out.append(l)
    return "\n".join(out) + "\n"


def partition_stream(row, first_ts_ms, last_ts_ms, extremes):
    """Records of one partition whose counters are the screenshot's input columns: `total` keyed live records,
    key / value lengths spread so that their sums are K-Bytes / V-Bytes exactly; `extremes` plants the
    750-byte and the 139-byte message (first partition only)."""
    i = row["inputs"]
    n, ksum, vsum = i["total"], i["key_size_sum"], i["value_size_sum"]
    assert i["tombstones"] == 0 and i["key_null"] == 0 and i["alive"] == n and i["key_non_null"] == n
    key_len = np.full(n, ksum // n, np.int32)
    key_len[:ksum % n] += 1
    val_len = np.empty(n, np.int32)
    fixed = 0
    if extremes:
        val_len[0] = G["expect"]["largest_message"] - key_len[0]
        val_len[1] = G["expect"]["smallest_message"] - key_len[1]
        fixed = 2
    rest = vsum - int(val_len[:fixed].sum())
    m = n - fixed
    val_len[fixed:] = rest // m
    val_len[fixed:fixed + rest % m] += 1
    ts = np.linspace(first_ts_ms, last_ts_ms, n).astype(np.int64)
    ts[0], ts[-1] = first_ts_ms, last_ts_ms
    return {"partition": np.full(n, row["partition"], np.int32), "key_len": key_len, "val_len": val_len,
            "ts_ms": ts}


def demo_streams():
    e = G["expect"]
    for k, row in enumerate(G["partitions"]):
        # +999 ms / +1 ms: the seconds are truncated (metric.rs:210), so neither moves the printed instant
        first = e["earliest_epoch_s"] * 1000 + 999 if k == 0 else e["earliest_epoch_s"] * 1000 + 5000 * k
        last = e["latest_epoch_s"] * 1000 + 1 if k == P - 1 else e["latest_epoch_s"] * 1000 - 7000 * (k + 1)
        yield row, partition_stream(row, first, last, extremes=(k == 0))


def input_vector():
    """The device counter-vector layout (kta_hip.h) filled with the screenshot's inputs."""
    v = np.zeros(P * 7 + 8, np.int64)
    c = v[:P * 7].reshape(P, 7)
    for r in G["partitions"]:
        i = r["inputs"]
        c[r["partition"]] = [i["total"], i["tombstones"], i["alive"], i["key_null"], i["key_non_null"],
                             i["key_size_sum"], i["value_size_sum"]]
    g, e = v[P * 7:], G["expect"]
    g[N.KTA_G_RECORDS] = e["overall_count"]
    g[N.KTA_G_NOT_MIN_TS_MS] = ~(e["earliest_epoch_s"] * 1000 + 999)
    g[N.KTA_G_MAX_TS_MS] = e["latest_epoch_s"] * 1000 + 1
    g[N.KTA_G_NOT_SMALLEST] = ~e["smallest_message"]
    g[N.KTA_G_LARGEST] = e["largest_message"]
    return v


def render(vec):
    lib = N.load()
    n, buf = C.c_size_t(), C.create_string_buffer(1 << 16)
    so = np.array([r["start_offset"] for r in G["partitions"]], np.int64)
    eo = np.array([r["end_offset"] for r in G["partitions"]], np.int64)
    rc = lib.kta_render_report(G["topic"].encode(), G["duration_secs"], vec.ctypes.data, P, 0, NOW[0], NOW[1],
                               so.ctypes.data, eo.ctypes.data, buf, len(buf), C.byref(n))
    assert rc == N.KTA_OK
    return buf.value.decode()


def check_accessors(mm):
    """`mm`: anything with the reference's accessor names (metric.rs:104-195)."""
    e = G["expect"]
    for r in G["partitions"]:
        p, x = r["partition"], r["expect"]
        assert mm.key_size_sum(p) + mm.value_size_sum(p) == x["p_bytes"]
        assert mm.key_size_avg(p) == x["key_size_avg"]
        assert mm.value_size_avg(p) == x["value_size_avg"]
        assert mm.message_size_avg(p) == x["message_size_avg"]
        assert oracle_py.format_f32_4(mm.dirty_ratio(p)) == x["dirty_ratio_4"]
    assert mm.overall_count() == e["overall_count"]
    assert mm.overall_count() // max(G["duration_secs"], 1) == e["msgs_per_sec"]
    assert mm.overall_size() == e["topic_size"]
    assert mm.largest_message() == e["largest_message"] and mm.smallest_message() == e["smallest_message"]


def test_python_oracle_accessors_and_report_reproduce_the_screenshot():
    """oracle_py's accessors + report printer from the screenshot's input columns: the text equals the
    screenshot character for character (with the screenshot build's two column names)."""
    mm = oracle_py.MessageMetrics(NOW)
    for r in G["partitions"]:
        p, i = r["partition"], r["inputs"]
        mm.total_messages[p], mm.tombstones_[p], mm.alive_[p] = i["total"], i["tombstones"], i["alive"]
        mm.key_non_null_[p], mm.key_size_sum_[p] = i["key_non_null"], i["key_size_sum"]
        mm.value_size_sum_[p] = i["value_size_sum"]  # key_null: no entry, as in the reference (never inc'ed)
    e = G["expect"]
    mm.overall_count_, mm.overall_size_ = e["overall_count"], e["topic_size"]
    mm.largest_message_, mm.smallest_message_ = e["largest_message"], e["smallest_message"]
    mm.earliest_message, mm.latest_message = (e["earliest_epoch_s"], 0), (e["latest_epoch_s"], 0)
    check_accessors(mm)
    so = {r["partition"]: r["start_offset"] for r in G["partitions"]}
    eo = {r["partition"]: r["end_offset"] for r in G["partitions"]}
    text = oracle_py.report(G["topic"], G["duration_secs"], mm, None, list(range(P)), so, eo, os_names=SCREEN_OS)
    assert text == "\n".join(G["text"]) + "\n"
    assert oracle_py.report(G["topic"], G["duration_secs"], mm, None, list(range(P)), so, eo) == to_v050(G["text"])


def test_product_decode_and_report_reproduce_the_screenshot():
    """kta_decode_vector + kta_render_report (the product's host side, no GPU needed) from the same inputs."""
    import kafka_topic_analyzer_amd as kta
    vec = input_vector()
    lib = N.load()
    res, counters = N.KtaResult(), np.zeros((P, 7), np.uint64)
    assert lib.kta_decode_vector(vec.ctypes.data, P, 0, C.byref(res), counters.ctypes.data) == N.KTA_OK
    mm = kta.MessageMetrics(res, counters, NOW)
    check_accessors(mm)
    assert mm.earliest_message() == (G["expect"]["earliest_epoch_s"], 0)
    assert mm.latest_message() == (G["expect"]["latest_epoch_s"], 0)
    assert render(vec) == to_v050(G["text"])


def test_c_oracle_reproduces_the_screenshot_from_a_stream_with_its_counters():
    """245 532 288 records through the C oracle's handle_message (metric.rs:207-252), one partition at a
    time as a Kafka consumer would deliver a topic read partition by partition."""
    o = Oracle(NOW)
    for _, cols in demo_streams():
        o.run_soa(cols)
    want = input_vector()
    assert np.array_equal(o.counters(P).astype(np.int64).ravel(), want[:P * 7])

    class Acc:  # the oracle's accessors under the reference's names
        def __getattr__(self, name):
            if name.endswith("_avg"):
                return lambda p: o.avg(name, p)
            return lambda *a: o.get(name, *a)
    check_accessors(Acc())
    assert o.earliest() == (G["expect"]["earliest_epoch_s"], 0)
    assert o.latest() == (G["expect"]["latest_epoch_s"], 0)


@pytest.mark.gpu
def test_gpu_path_reproduces_the_screenshot_from_a_stream_with_its_counters():
    """The same stream through the C ABI on the GPU (pinned staging -> scan kernel -> fold -> kta_finish),
    then the product's report printer: the screenshot's text (v0.5.0 column names)."""
    import kafka_topic_analyzer_amd as kta
    with kta.HipMetricHandler(P, count_alive_keys=False, device=0, batch_capacity=1 << 22, now=NOW) as h:
        for _, cols in demo_streams():
            h.submit_columns(cols["partition"], cols["key_len"], cols["val_len"], cols["ts_ms"])
        res, counters = h.finish()
        vec = h.result_vector_host().astype(np.int64)
        mm = kta.MessageMetrics(res, counters, NOW)
    assert np.array_equal(vec, input_vector())
    check_accessors(mm)
    assert render(vec) == to_v050(G["text"])
