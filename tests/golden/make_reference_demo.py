"""Writes tests/golden/reference_demo_output.json: a transcription of /root/reference/demo_output.png
(linked from /root/reference/README.md:28), the one output of the reference itself that the reference ships.

The screenshot is a real run of the reference (an older build: its legend and header say `|< OS` / `>| OS`
where src/main.rs:150,175 of v0.5.0 say `< OS` / `> OS`) against a 10-partition topic.  The numbers below
were read off the image by hand; the screenshot's own redundancy checks the reading:
    sum(Total)   = 245 532 288, // 416 s = 590 221        = "Estimated Msg/s"   (main.rs:130)
    sum(P-Bytes) = 66 434 997 213                          = "Topic Size"        (main.rs:137)
    P-Bytes      = K-Bytes + V-Bytes on every row                                (main.rs:165)
and this script refuses to write the file unless they hold.  Inputs of the reference's accessors are the
columns Total / Alive / Tmb / K Null / K !Null / K-Bytes / V-Bytes (the seven counter maps of
metric.rs:12-18); everything else on the screen is an OUTPUT of metric.rs:132-183 / main.rs:125-178 and is
what the tests assert.

Run: python tests/golden/make_reference_demo.py   (needs nothing but the standard library)."""
import calendar
import json
import os

TOPIC = "global.trv_bulk.partner_import"
SECS = 416
MSGS_PER_SEC = 590221
EARLIEST = "2018-01-31 17:23:13 UTC"
LATEST = "2018-04-13 14:29:52 UTC"
LARGEST, SMALLEST, TOPIC_SIZE = 750, 139, 66434997213
HEADER = ["P", "|< OS", ">| OS", "Total", "Alive", "Tmb", "DR", "K Null", "K !Null", "P-Bytes", "K-Bytes",
          "V-Bytes", "A K-Sz", "A V-Sz", "A M-Sz"]
# P, |< OS, >| OS, Total, Alive, Tmb, DR, K Null, K !Null, P-Bytes, K-Bytes, V-Bytes, A K-Sz, A V-Sz, A M-Sz
ROWS = """
0 0 112298537 25056009 25056009 0 0.0000 0 25056009 6778805354 225504081 6553301273 9 261 270
1 0 112244988 25063295 25063295 0 0.0000 0 25063295 6780199421 225569655 6554629766 9 261 270
2 0 112295570 25056714 25056714 0 0.0000 0 25056714 6777635839 225510426 6552125413 9 261 270
3 0 112275362 25058243 25058243 0 0.0000 0 25058243 6778031556 225524187 6552507369 9 261 270
4 0 112315450 25062939 25062939 0 0.0000 0 25062939 6780416185 225566451 6554849734 9 261 270
5 0 112267563 25063360 25063360 0 0.0000 0 25063360 6779370776 225570240 6553800536 9 261 270
6 0 112262485 25043793 25043793 0 0.0000 0 25043793 6774475467 225394137 6549081330 9 261 270
7 0 112147975 25038860 25038860 0 0.0000 0 25038860 6772769509 225349740 6547419769 9 261 270
8 0 112332976 20021871 20021871 0 0.0000 0 20021871 5432377054 180196839 5252180215 9 262 271
9 0 112279184 25067204 25067204 0 0.0000 0 25067204 6780916052 225604836 6555311216 9 261 270
"""


def table_text(rows):
    """The table as the screenshot shows it: prettytable-rs default format (main.rs:149-176)."""
    w = [max(len(r[i]) for r in rows) for i in range(len(rows[0]))]
    sep = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    out = [sep]
    for r in rows:
        out.append("|" + "|".join(" " + c.ljust(w[i]) + " " for i, c in enumerate(r)) + "|")
        out.append(sep)
    return out


def epoch(s):
    d, t, _ = s.split(" ")
    y, mo, da = map(int, d.split("-"))
    h, mi, se = map(int, t.split(":"))
    return calendar.timegm((y, mo, da, h, mi, se, 0, 0, 0))


def main():
    rows = [l.split() for l in ROWS.strip().split("\n")]
    parts = []
    for r in rows:
        p = dict(partition=int(r[0]), start_offset=int(r[1]), end_offset=int(r[2]),
                 inputs=dict(total=int(r[3]), alive=int(r[4]), tombstones=int(r[5]), key_null=int(r[7]),
                             key_non_null=int(r[8]), key_size_sum=int(r[10]), value_size_sum=int(r[11])),
                 expect=dict(dirty_ratio_4=r[6], p_bytes=int(r[9]), key_size_avg=int(r[12]),
                             value_size_avg=int(r[13]), message_size_avg=int(r[14])))
        assert p["expect"]["p_bytes"] == p["inputs"]["key_size_sum"] + p["inputs"]["value_size_sum"], r
        parts.append(p)
    assert sum(p["inputs"]["total"] for p in parts) // SECS == MSGS_PER_SEC
    assert sum(p["expect"]["p_bytes"] for p in parts) == TOPIC_SIZE
    text = ["", "=" * 120, "Calculating statistics...", "Topic " + TOPIC, "Scanning took: %d seconds" % SECS,
            "Estimated Msg/s: %d" % MSGS_PER_SEC, "-" * 120, "Earliest Message: " + EARLIEST,
            "Latest Message: " + LATEST, "-" * 120, "Largest Message: %d bytes" % LARGEST,
            "Smallest Message: %d bytes" % SMALLEST, "Topic Size: %d bytes" % TOPIC_SIZE, "=" * 120,
            "| K = Key, V = Value, P = Partition, Tmb = Tombstone(s), Sz = Size",
            "| DR = Dirty Ratio, A = Average, Lst = last, |< OS = start offset, >| OS = end offset"]
    text += table_text([HEADER] + rows) + ["", "=" * 120]
    doc = dict(
        source="/root/reference/demo_output.png (README.md:28), transcribed by hand; see make_reference_demo.py",
        note="screenshot build prints '|< OS' / '>| OS'; src/main.rs:150,175 (v0.5.0) print '< OS' / '> OS'",
        topic=TOPIC, duration_secs=SECS, count_alive_keys=False,
        expect=dict(msgs_per_sec=MSGS_PER_SEC, earliest=EARLIEST, latest=LATEST, earliest_epoch_s=epoch(EARLIEST),
                    latest_epoch_s=epoch(LATEST), largest_message=LARGEST, smallest_message=SMALLEST,
                    topic_size=TOPIC_SIZE, overall_count=sum(p["inputs"]["total"] for p in parts)),
        header_screenshot=HEADER, partitions=parts, text=text)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_demo_output.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
