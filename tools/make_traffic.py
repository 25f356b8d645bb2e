"""profiles/traffic.json from the PMC summary of tools/profile_round.sh (profiles/rNN_pmc_hbm.txt): HBM bytes per
launch of the kernels bench.py reports a roofline for.  bench.py replays this file (roofline.traffic); it
does not measure traffic itself.  Every entry records the kernel's source file and its sha256 (first 16 hex
digits) AS THE FILE LIES IN THE TREE WHEN THIS SCRIPT RUNS — run it on the tree the passes were taken with;
bench.py returns "traffic": null for a kernel whose file has changed since.

    python tools/make_traffic.py profiles/r05_pmc_hbm.txt > profiles/traffic.json

Counters are KiB per launch.  FETCH_SIZE under-reports wide coalesced streaming reads by a factor of two on
gfx950 (MI355X_MICROARCH.md, HBM section), so the streaming part of a kernel's reads is doubled:
  kta_metrics_scan, kafka_decode_coop                        everything they read is such a stream
  kta_alive_partition32                                       reads its columns 4 bytes per lane (256-byte requests of a wave)
                                                              and the keys 16 bytes per lane: whether the factor of two
                                                              applies is decided by the cross-check below (28 B per record)
  kta_alive_apply (bit set state)                             the pair stream (4 B x records) and the bucket regions of
                                                              the bit set (16 B per lane, 2 KiB per wave) both are
Cross-checks the corrections must pass (printed to stderr): the scan reads 20 B/record, the partition kernel
28 B/record.
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src(rel):
    """rel: one file, or the files a kernel lives in joined by "+" (bench.py: _source_hash reads it the same way)."""
    digests = [hashlib.sha256(open(os.path.join(ROOT, r), "rb").read()).hexdigest() for r in rel.split("+")]
    return {"source_file": rel,
            "source_sha256_16": (digests[0] if len(digests) == 1 else hashlib.sha256("".join(digests).encode()).hexdigest())[:16]}

path = sys.argv[1]
rows = {}          # (kernel prefix, counter) -> (n, avg, min, max)
for line in open(path):
    m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m:
        rows[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)))


def find(sub, counter, largest_grid=False):
    hits = [(k, v) for (k, c), v in rows.items() if c == counter and sub in k]
    if largest_grid and len(hits) > 1:         # rows are per (kernel, grid size): the biggest launch
        hits.sort(key=lambda kv: int(re.search(r"grid=(\d+)", kv[0]).group(1)))
        hits = hits[-1:]
    assert len(hits) == 1, (sub, counter, [k for k, _ in hits])
    return hits[0][1]


KIB = 1024.0
out = {"round": 6, "source": path,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 5 --warmup 1 "
                 "--preroll 5 --no-cpu-baseline` (tools/profile_round.sh; --no-alive-extras: every kernel at one launch size), turned into this file by tools/make_traffic.py; counters "
                 "are KiB per launch (average over the launches of the kernel unless stated).  FETCH_SIZE is doubled for wide "
                 "coalesced streaming reads per the gfx950 correction (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported.  "
                 "bench.py replays these numbers (roofline.traffic, traffic_source), it does not measure them."}

n_scan = 1 << 30
f, w = find("kta_metrics_scan<0, true, false>", "FETCH_SIZE"), find("kta_metrics_scan<0, true, false>", "WRITE_SIZE")
rd, wr = 2 * f[1] * KIB, w[1] * KIB
out["kta_metrics_scan"] = {"kernel": "kta_metrics_scan<0,true,false>", "records_per_launch": n_scan,
                           "algorithmic_bytes_per_launch": 20 * n_scan, "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1],
                           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                           "ratio_to_algorithmic": (rd + wr) / (20 * n_scan), **src("kafka_topic_analyzer_amd/csrc/kta_kernels.hip")}
print("scan: read %.3f GB vs 20 B x 2^30 = %.3f GB" % (rd / 1e9, 20 * n_scan / 1e9), file=sys.stderr)

n_alive = 15 << 24                            # bench.py --alive-records
ALIVE_SRC = src("kafka_topic_analyzer_amd/csrc/kta_alive.hip")


def entry(key, kernel, match, algo, fetch_factor, note, source=ALIVE_SRC, n=n_alive):
    f, w = find(match, "FETCH_SIZE"), find(match, "WRITE_SIZE")
    rd, wr = fetch_factor * f[1] * KIB, w[1] * KIB
    out[key] = {"kernel": kernel, "records_per_launch": n, "algorithmic_bytes_per_launch": algo, "launches_fetch_pass": f[0],
                "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1], "FETCH_SIZE_factor": fetch_factor,
                "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "note": note, **source}
    return rd, wr


# The factor of FETCH_SIZE is 2 for these kernels as for the scan: pass 1 reads its columns 4 bytes per lane and the keys 16
# bytes per lane, and FETCH_SIZE x 2 lands on the 28 (fused: 40) bytes per record it is known to read — the cross-check below.
rd, wr = entry("kta_alive_partition32", "kta_alive_partition32<10,false>", "kta_alive_partition32<10, false>", 28 * n_alive, 2.0,
               "reads = the batch (key_len, val_len, key_off, 16 B keys: 28 B/record); writes = the partitioned 4-byte pairs (slot in "
               "bucket, window, alive), one per record that survives the guard, in aligned 64-byte blocks")
print("partition32: read %.3f GB vs 28 B x %d = %.3f GB; wrote %.3f GB vs 4 B x records = %.3f GB"
      % (rd / 1e9, n_alive, 28 * n_alive / 1e9, wr / 1e9, 4 * n_alive / 1e9), file=sys.stderr)
rd, wr = entry("kta_alive_partition32_fused", "kta_alive_partition32<10,true>", "kta_alive_partition32<10, true>", 40 * n_alive, 2.0,
               "both handlers in one pass: reads = partition, key_len, val_len, ts_ms, key_off, 16 B keys (40 B/record); writes = the "
               "4-byte pairs + one row of the scan's partial workspace per workgroup")
print("partition32 fused: read %.3f GB vs 40 B x %d = %.3f GB; wrote %.3f GB" % (rd / 1e9, n_alive, 40 * n_alive / 1e9, wr / 1e9), file=sys.stderr)
rd, wr = entry("kta_alive_apply", "kta_alive_apply<10,true,false>", "kta_alive_apply<10, true, false>", 0, 2.0,
               "reads = the pair stream (4 B x records) + the 512 MiB bit set, both wide coalesced streams (FETCH_SIZE doubled); "
               "writes = the 512 MiB bit set in whole lines.  None of this is algorithmic input: the batch's algorithmic bytes are "
               "booked on kta_alive_partition32")
print("apply: read %.3f GB vs pairs %.3f + bit set 0.537 GB; wrote %.3f GB vs 0.537" % (rd / 1e9, 4 * n_alive / 1e9, wr / 1e9), file=sys.stderr)
# table state (alive_pass_table / both_handlers_table): pass 1 with 6-byte pairs (the seq column's order checked by the consumer
# waves), alone and with the metrics handler's work in it; pass 2 on the 32 GiB table
rd, wr = entry("kta_alive_partition48", "kta_alive_partition48<10,true,false>", "kta_alive_partition48<10, true, false", 36 * n_alive, 2.0,
               "table state, batches with a seq column: reads = the batch (28 B/record) + the seq column (8 B/record, read by the consumer "
               "waves, every line once); writes = 6-byte pairs (slot in the bucket, alive, index inside the workgroup's range) as a dense "
               "stream in 64-byte blocks")
print("partition48 (table state): read %.3f GB vs 36 B x records = %.3f GB; wrote %.3f GB vs 6 B x records = %.3f GB"
      % (rd / 1e9, 36 * n_alive / 1e9, wr / 1e9, 6 * n_alive / 1e9), file=sys.stderr)
rd, wr = entry("kta_alive_partition48_fused", "kta_alive_partition48<10,true,true>", "kta_alive_partition48<10, true, true", 48 * n_alive, 2.0,
               "table state, both handlers in the one pass: reads = partition, key_len, val_len, ts_ms, key_off, 16 B keys (40 B/record) + "
               "the seq column (8 B/record); writes = the 6-byte pairs + one row of the scan's partial workspace per workgroup")
print("partition48 fused: read %.3f GB vs 48 B x records = %.3f GB; wrote %.3f GB" % (rd / 1e9, 48 * n_alive / 1e9, wr / 1e9), file=sys.stderr)
rd, wr = entry("kta_alive_apply_table", "kta_alive_apply<10,false,false>", "kta_alive_apply<10, false, false>", 0, 2.0,
               "table state: reads = the 6-byte pair stream (doubled: a wide stream) + one 8-byte table entry and one seq value per "
               "surviving slot (the doubling overstates those scattered reads); writes = one partial write per slot whose entry changes "
               "+ the written list")
print("apply (table state): read %.3f GB vs pairs %.3f GB + survivors; wrote %.3f GB" % (rd / 1e9, 6 * n_alive / 1e9, wr / 1e9), file=sys.stderr)

# the 4 M-record launch of ~16 KiB batches: 66 667 batches x 16 lanes -> grid 1 066 688; batches below 28 KiB take <4, 3 KiB, 16>
# (kta_kafka.hip: pick_geometry) — the row is found by that grid size, whichever geometry served it
DECODE_GRID = 1066688
hits = sorted({k for (k, c) in rows if "kafka_decode_coop<" in k and "grid=%d" % DECODE_GRID in k})
assert len(hits) == 1, hits
decode_kernel = re.search(r"kafka_decode_coop<[^>]*>", hits[0]).group(0)
f, w = rows[(hits[0], "FETCH_SIZE")], rows[(hits[0], "WRITE_SIZE")]
raw_log = 1075251127                          # bytes of the 4 M-record raw log bench.py's kafka_decode.roofline describes
rd, wr = 2 * f[1] * KIB, w[1] * KIB
out["kafka_decode_coop"] = {"kernel": decode_kernel, "records_per_launch": 4000000,
                            "algorithmic_bytes_per_launch": raw_log, "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1],
                            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                            "ratio_to_algorithmic": (rd + wr) / raw_log,
                            "note": "rows are per launch size (the summary carries grid sizes): grid %d = the 4 M-record / "
                                    "1.075 GB launches bench.py's kafka_decode.roofline describes, %d of them" % (DECODE_GRID, f[0]),
                            **src("kafka_topic_analyzer_amd/csrc/kta_kafka.hip+kafka_topic_analyzer_amd/csrc/kta_decode_coop.h+"
                                  "kafka_topic_analyzer_amd/csrc/kta_records.h")}
print("decode: read %.3f GB + wrote %.3f GB vs raw log %.3f GB" % (rd / 1e9, wr / 1e9, raw_log / 1e9), file=sys.stderr)
json.dump(out, sys.stdout, indent=1)
print()
