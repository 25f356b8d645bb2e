"""profiles/traffic.json from the PMC summary of tools/profile_r02.sh (profiles/rNN_pmc_hbm.txt): HBM bytes per
launch of the kernels bench.py reports a roofline for.  bench.py replays this file (roofline.traffic); it
does not measure traffic itself.

    python tools/make_traffic.py profiles/r02_pmc_hbm.txt > profiles/traffic.json

Counters are KiB per launch.  FETCH_SIZE under-reports wide coalesced streaming reads by a factor of two on
gfx950 (MI355X_MICROARCH.md, HBM section), so the streaming part of a kernel's reads is doubled:
  kta_metrics_scan, kta_alive_partition, kafka_decode_coop   everything they read is such a stream
  kta_alive_apply                                             the pair stream (8 B x records) is; the one 8-byte
                                                              agent-scope read per surviving slot is not
Cross-checks the corrections must pass (printed to stderr): the scan reads 20 B/record, the partition kernel
28 B/record.
"""
import json
import re
import sys

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
out = {"round": 2, "source": path,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 5 --warmup 1 "
                 "--preroll 5 --no-cpu-baseline` (tools/profile_r02.sh), turned into this file by tools/make_traffic.py; counters "
                 "are KiB per launch (average over the launches of the kernel unless stated).  FETCH_SIZE is doubled for wide "
                 "coalesced streaming reads per the gfx950 correction (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported.  "
                 "bench.py replays these numbers (roofline.traffic, traffic_source), it does not measure them."}

n_scan = 1 << 30
f, w = find("kta_metrics_scan<0, true, false>", "FETCH_SIZE"), find("kta_metrics_scan<0, true, false>", "WRITE_SIZE")
rd, wr = 2 * f[1] * KIB, w[1] * KIB
out["kta_metrics_scan"] = {"kernel": "kta_metrics_scan<0,true,false>", "records_per_launch": n_scan,
                           "algorithmic_bytes_per_launch": 20 * n_scan, "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1],
                           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                           "ratio_to_algorithmic": (rd + wr) / (20 * n_scan)}
print("scan: read %.3f GB vs 20 B x 2^30 = %.3f GB" % (rd / 1e9, 20 * n_scan / 1e9), file=sys.stderr)

n_alive = 1 << 26
f, w = find("kta_alive_partition<10>", "FETCH_SIZE"), find("kta_alive_partition<10>", "WRITE_SIZE")
rd, wr = 2 * f[1] * KIB, w[1] * KIB
out["kta_alive_partition"] = {"kernel": "kta_alive_partition<10>", "records_per_launch": n_alive,
                              "algorithmic_bytes_per_launch": 28 * n_alive, "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1],
                              "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                              "note": "reads = the batch (key_len, val_len, key_off, 16 B keys: 28 B/record); writes = the partitioned "
                                      "(hash, sequence, alive) pairs, 8 B per keyed record in aligned 64-byte blocks"}
print("partition: read %.3f GB vs 28 B x 2^26 = %.3f GB" % (rd / 1e9, 28 * n_alive / 1e9), file=sys.stderr)

f, w = find("kta_alive_apply<10, 14>", "FETCH_SIZE"), find("kta_alive_apply<10, 14>", "WRITE_SIZE")
pairs = 8.0 * n_alive                         # the coalesced pair stream: reported at half, so half of it is added back
rd, wr = f[1] * KIB + pairs / 2, w[1] * KIB
out["kta_alive_apply"] = {"kernel": "kta_alive_apply<10,14>", "records_per_launch": n_alive, "algorithmic_bytes_per_launch": 0,
                          "FETCH_SIZE_kib_avg": f[1], "WRITE_SIZE_kib_avg": w[1], "hbm_read_bytes_per_launch": rd,
                          "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                          "note": "reads = the pair stream (8 B x 2^26, a 16 B/lane coalesced stream: reported at half, so half of it "
                                  "is added back) + one 8-byte agent-scope read per surviving slot (counted at the size of the memory "
                                  "request); writes = one 8-byte agent-scope store per surviving slot (a partial write of a 64-byte "
                                  "block).  None of this is algorithmic input: the batch's algorithmic bytes are booked on "
                                  "kta_alive_partition"}

f, w = find("kafka_decode_coop<4, 2048u, 32u>", "FETCH_SIZE", True), find("kafka_decode_coop<4, 2048u, 32u>", "WRITE_SIZE", True)
raw_log = 1075251127                          # bytes of the 4 M-record raw log bench.py's kafka_decode.roofline describes
rd, wr = 2 * f[3] * KIB, w[3] * KIB
out["kafka_decode_coop"] = {"kernel": "kafka_decode_coop<4, 2048u, 32u>", "records_per_launch": 4000000,
                            "algorithmic_bytes_per_launch": raw_log, "FETCH_SIZE_kib_max": f[3], "WRITE_SIZE_kib_max": w[3],
                            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                            "ratio_to_algorithmic": (rd + wr) / raw_log,
                            "note": "rows are per launch size where the summary carries grid sizes (the largest grid = the 4 M-record / "
                                    "1.075 GB launches bench.py's kafka_decode.roofline describes, %d of them); the MAX of the row is "
                                    "used, which also picks that launch out of a mixed row" % f[0]}
json.dump(out, sys.stdout, indent=1)
print()
