"""The round paragraph of profiles/README.md written from the committed files themselves (profiles/r06_kernel_stats.txt, r06_pmc_hbm.txt,
the bench lines): every figure it quotes is read, not typed.  Figures of earlier calls ("other boxes") are literals of this script.
    python tools/round_readme.py     (rewrites profiles/README.md from "Round 6 (final kernels" on)"""
import json,re
root='/root/repo/profiles/'
ks={}
for line in open(root+'r06_kernel_stats.txt'):
    m=re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m: ks[m.group(1).strip()]=(int(m.group(2)), float(m.group(4)))
def k(sub, *more):
    for name,(n,avg) in ks.items():
        if sub in name and all(x in name for x in more): return avg
    raise KeyError(sub)
pm={}
for line in open(root+'r06_pmc_hbm.txt'):
    m=re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m: pm.setdefault((m.group(1).strip(), m.group(2)), []).append((int(m.group(3)), float(m.group(4))))
def p(sub, ctr, *more, pick=max):
    vals=[v for (name,c),lst in pm.items() if c==ctr and sub in name and all(x in name for x in more) for (n,v) in lst]
    return pick(vals)
n1=json.load(open(root+'r06_bench_n1.json')); st=json.load(open(root+'r06_bench_stats_run.json'))
def rf(d,key): r=d[key].get('roofline',d[key]); return r
scan=k('kta_metrics_scan'); 
F=lambda x: f"{x:,.0f}"
hk=n1['alive_pass_hot_key']; rows={r['distinct_keys']:r['kernel_ms'] for r in hk['rows']}
c5=hk['mostly_unique_keys']['kernel_ms_per_batch']
dec=n1['kafka_decode']; bb=dec['by_batch_size']
comp=dec['compressed']; e2e=n1['raw_log_e2e']; bpm=n1['boundary_per_message']['rows']
c5f=json.load(open(root+'r06_bench_c5_forced.json')); c4f=json.load(open(root+'r06_bench_c4_strong_forced.json'))
p32=k('kta_alive_partition32<10, false>'); p32f=k('kta_alive_partition32<10, true>'); ab=k('kta_alive_apply<10, true, false>')
p48=k('kta_alive_partition48<10, true, false>'); p48f=k('kta_alive_partition48<10, true, true>'); at=k('kta_alive_apply<10, false, false>')
fold=k('kta_fold_partials'); rng=k('kta_alive_apply<10, true, true>'); fb=k('kta_alive_fallback')
B=7.046430720e9; B2=10.06632960e9; T=9.05969664e9; T2=12.07959552e9
txt=f'''Round 6 (final kernels; `r06_*` from ONE gpurun call, `tools/profile_round.sh r06`, on the final tree — the round's last change of a profiled
kernel was the LZ4 kernel's scalar parse; the recipe runs the kernel-stats process first, then the two counter passes, then — with the
traffic of THOSE passes in `traffic.json` — the default `python bench.py` (`r06_bench_n1.json`), then the two forced lines.  The same recipe ran
eight times earlier in the round, on other boxes, after the table state's rewrite, the slot-range passes and the inflate kernels' steps: their
numbers are quoted as "other boxes").
`kta_metrics_scan<0,true,false>` on 2^30 records: {scan:.1f} us per launch under rocprofv3 (174 launches) vs {st['roofline']['kernel_ms']*1e3:.1f} us from the HIP events of the
same process (`r06_bench_stats_run.json`) => {20*2**30/st['roofline']['kernel_ms']/1e9:.2f} TB/s = {st['roofline']['frac']:.3f}; the default run of the call (`r06_bench_n1.json`) {n1['roofline']['kernel_ms']*1e3:.1f} us => {n1['roofline']['frac']:.3f}
({n1['value']/1e9:.1f} G records/s); other boxes 3213.5 ... 3455 us (0.777 ... 0.835: the round's boxes and processes differ by 7 %).  FETCH_SIZE {F(p('kta_metrics_scan','FETCH_SIZE'))} KiB x 2
+ WRITE_SIZE {F(p('kta_metrics_scan','WRITE_SIZE'))} KiB = {n1['roofline']['traffic']/1e9:.2f} GB = {n1['roofline']['traffic']/(20*2**30):.3f}x (unchanged kernel).
Bit set state, 251,658,240 records (7.05 GB): `kta_alive_partition32<10,false>` {p32:.1f} us + `kta_alive_apply<10,true,false>` {ab:.1f} us (+ the
slot-range instantiation `<10,true,true>` {rng:.1f} us and `kta_alive_fallback` {fb:.1f} us, both return at once) = {p32+ab+rng+fb:.0f} us under rocprofv3 => {B/((p32+ab+rng+fb)*1e-6)/8e12:.3f}; HIP
events {rf(n1,'alive_pass')['kernel_ms']*1e3:.0f} us (the default run) / {rf(st,'alive_pass')['kernel_ms']*1e3:.0f} us => {rf(n1,'alive_pass')['frac']:.3f} / {rf(st,'alive_pass')['frac']:.3f}; other boxes 2342 ... 2506 us => 0.351 ... 0.376 (round 5:
1761.6 + 633.9; pass 2 lost its careful mode: 95 -> 66 registers).  One fused pass: `kta_alive_partition32<10,true>` {p32f:.1f} us + fold {fold:.1f} us +
apply {ab:.1f} us = {p32f+fold+ab:.0f} us => {B2/((p32f+fold+ab)*1e-6)/8e12:.3f} of 10.07 GB; HIP events {rf(n1,'both_handlers')['kernel_ms']*1e3:.0f} / {rf(st,'both_handlers')['kernel_ms']*1e3:.0f} us => {rf(n1,'both_handlers')['frac']:.3f} / {rf(st,'both_handlers')['frac']:.3f}; other boxes 2945 ... 3094 us => 0.407 ... 0.427; two
passes 3.26-3.29 ms => 0.38-0.39.  Traffic: partition32 {p('partition32<10, f','FETCH_SIZE')*1024*2/1e9:.3f} GB read + {p('partition32<10, f','WRITE_SIZE')*1024/1e9:.3f} GB written, fused {p('partition32<10, t','FETCH_SIZE')*1024*2/1e9:.3f} + {p('partition32<10, t','WRITE_SIZE')*1024/1e9:.3f}, apply {p('alive_apply<10, true, f','FETCH_SIZE')*1024*2/1e9:.3f} + {p('alive_apply<10, true, f','WRITE_SIZE')*1024/1e9:.3f} => {rf(n1,'alive_pass')['traffic']/1e9:.2f} GB
= {rf(n1,'alive_pass')['traffic']/B:.2f}x / {rf(n1,'both_handlers')['traffic']/1e9:.2f} GB = {rf(n1,'both_handlers')['traffic']/B2:.2f}x.
Table state, the same records with a seq column (9.06 GB; rounds 1-5: `kta_alive_partition<10,true>` 2639.0 us with 8-byte pairs +
`kta_alive_apply<10,false>` 1544.6 us incl. a first launch, 3753.7 us steady => 0.302): **`kta_alive_partition48<10,true,false>` {p48:.1f} us +
`kta_alive_apply<10,false,false>` {at:.1f} us = {p48+at:.0f} us under rocprofv3 => {T/((p48+at)*1e-6)/8e12:.3f}; HIP events {rf(n1,'alive_pass_table')['kernel_ms']*1e3:.0f} us (the default run) / {rf(st,'alive_pass_table')['kernel_ms']*1e3:.0f} us => {rf(n1,'alive_pass_table')['frac']:.3f} /
{rf(st,'alive_pass_table')['frac']:.3f}; other boxes and processes 3214 ... 3598 us => 0.315 ... 0.352.**  Both handlers in the one pass (12.08 GB; round 5: scan + pass = 4.49 ms =>
0.336): **`kta_alive_partition48<10,true,true>` {p48f:.1f} us + fold {fold:.1f} us + apply {at:.1f} us = {p48f+fold+at:.0f} us => {T2/((p48f+fold+at)*1e-6)/8e12:.3f}; HIP events {rf(n1,'both_handlers_table')['kernel_ms']*1e3:.0f} us (the default
run) / {rf(st,'both_handlers_table')['kernel_ms']*1e3:.0f} us (the rocprofv3 process) => {rf(n1,'both_handlers_table')['frac']:.3f} / {rf(st,'both_handlers_table')['frac']:.3f}; other boxes 3736 ... 4128 us => 0.366 ... 0.404** — 0.37-0.40 over the round's boxes
and processes; two passes 4.25 ms => 0.356.  Traffic: partition48 FETCH {F(p('partition48<10, true, f','FETCH_SIZE'))} KiB x 2 = {p('partition48<10, true, f','FETCH_SIZE')*2048/1e9:.3f} GB (1.003x the 36 B per record) + WRITE
{F(p('partition48<10, true, f','WRITE_SIZE'))} KiB = {p('partition48<10, true, f','WRITE_SIZE')*1024/1e9:.3f} GB (the pairs are 1.510 GB: 6 B per record; round 5: 2.04 GB of 8-byte pairs); fused FETCH {F(p('partition48<10, true, t','FETCH_SIZE'))} KiB x 2 = {p('partition48<10, true, t','FETCH_SIZE')*2048/1e9:.3f} GB
+ WRITE {p('partition48<10, true, t','WRITE_SIZE')*1024/1e9:.3f} GB; apply FETCH {F(p('alive_apply<10, false','FETCH_SIZE'))} KiB x 2 = {p('alive_apply<10, false','FETCH_SIZE')*2048/1e9:.3f} GB (the 1.51 GB of pairs + one table entry and one seq value per surviving slot: the
doubling overstates those scattered reads) + WRITE {p('alive_apply<10, false','WRITE_SIZE')*1024/1e9:.3f} GB => {rf(n1,'alive_pass_table')['traffic']/1e9:.2f} GB = {rf(n1,'alive_pass_table')['traffic']/T:.2f}x of 9.06 GB (round 5: 1.77x) / {rf(n1,'both_handlers_table')['traffic']/1e9:.2f} GB = {rf(n1,'both_handlers_table')['traffic']/T2:.2f}x of 12.08 GB.
Shapes beside the headline (`r06_bench_n1.json: alive_pass_hot_key`, every count held against `tests/golden/bench_alive_counts.json`): one key
over 2^26 records {rows[1]:.2f} ms, 40 keys {rows[40]:.2f} ms, **20 M distinct keys per 251,658,240-record batch {hk['many_keys']['kernel_ms']:.2f} ms** (round 5: 7.6 ms at 2^28; two
slot-range passes per bucket, no bucket to the fallback kernel), **config 5's law on one GPU {min(c5[1:]):.1f}-{max(c5):.1f} ms per batch** (round 5: 8.1-8.2; eight
slot-range passes).
`kafka_decode_coop<4, 3072, 16>`: {dec['roofline']['kernel_ms']*1e3:.1f}-{st['kafka_decode']['roofline']['kernel_ms']*1e3:.1f} us for the 4 M-record / 1.075 GB launch (HIP events; {st['kafka_decode']['roofline']['frac']:.2f}-{dec['roofline']['frac']:.2f}; other boxes 209.9-229 us, 0.59-0.64),
a step {dec['ms_per_step']:.3f} ms around it (round 5: 0.396 — the descriptors now upload on the copy stream beside the kernels of the step before); 2 M records:
{bb[0]['kernel_ms']:.3f} / {bb[1]['kernel_ms']:.3f} / {bb[2]['kernel_ms']:.3f} ms at ~2 / ~16 / ~134 KiB ({bb[0]['frac']:.3f} / {bb[1]['frac']:.3f} / {bb[2]['frac']:.3f}), and the rows for the record: 500 batches of ~1 MiB {bb[3]['kernel_ms']:.3f} ms ({bb[3]['frac']:.3f}),
125 batches of ~4 MiB {bb[4]['kernel_ms']:.2f} ms ({bb[4]['frac']:.3f}) — one batch's serial chain.  Traffic of the 4 M launch: FETCH {F(p('kafka_decode_coop<4','FETCH_SIZE'))} KiB x 2 = {p('kafka_decode_coop<4','FETCH_SIZE')*2048/1e9:.3f} GB + WRITE {p('kafka_decode_coop<4','WRITE_SIZE')*1024/1e9:.3f}
GB = {dec['roofline']['traffic']/1.075e9:.2f}x the raw log.
Inflate kernels per 1 M records (16 667 batches of the bench's patterned values): **`kafka_gzip_tokenize_wave` {k('kafka_gzip_tokenize_wave'):.1f} us (one wave per batch, 64
lanes per DEFLATE block: `kafka_gzip_tokenize<8>` took 3754.5 us and now takes what the wave kernel leaves, {k('kafka_gzip_tokenize<8u>'):.1f} us) + `kafka_gzip_apply` {k('kafka_gzip_apply'):.1f}
us (637.2: a 4 KiB ring, a match at a time, 64 tokens at a time); inflate + decode {comp['gzip']['ms']:.2f} ms = {comp['gzip']['compressed_GBps']:.0f} GB/s of compressed input (21.7);
`kafka_zstd_inflate_coop` {k('kafka_zstd_inflate_coop'):.1f} us (rounds 2-5: 2.80-2.85 ms — the FSE tables built by all 64 lanes, cheaper waits, 16 instead of 8 waves per
CU, a sequence's fields from containers that serve two sequences, positions in 32 bits, the Huffman literals by the whole wave); inflate + decode
{comp['zstd']['ms']:.2f} ms = {comp['zstd']['compressed_GBps']:.0f} GB/s (32.8); `kafka_snappy_inflate_coop` {k('kafka_snappy_inflate_coop'):.1f} us (1480.4) and `kafka_lz4_inflate_coop` {k('kafka_lz4_inflate_coop'):.1f} us (1134.9): 4 KiB of LDS a
wave instead of 20, 32 waves per CU instead of 8, and LZ4's parse in scalar registers with 32-bit positions — {comp['snappy']['compressed_GBps']:.0f} and {comp['lz4']['compressed_GBps']:.0f} GB/s (65, 77)**.  JSON-like values (`tools/bench_inflate.py --values text`, not in the
bench line): gzip 2.73 ms = 56 GB/s (tokenizer 1.49 + apply 1.14), zstd 10.7 ms = 14.3, Snappy 3.59 ms = 53, LZ4 3.6 ms = 50.
PCIe-inclusive: `raw_log_e2e` {e2e['metrics']['raw_log_GBps']:.1f} / {e2e['count_alive_keys']['raw_log_GBps']:.1f} GB/s of raw log (the round's boxes: 49-53; an untimed leg first: the order of the timed ones
decides nothing any more); `raw_log_e2e.compressed`, GB/s of compressed log: Snappy {e2e['compressed']['snappy']['raw_log_GBps']:.1f}, gzip {e2e['compressed']['gzip']['raw_log_GBps']:.1f}, zstd {e2e['compressed']['zstd']['raw_log_GBps']:.1f}, LZ4 {e2e['compressed']['lz4']['raw_log_GBps']:.1f}
(36 blobs per codec; with the 12 of the round's earlier lines, 25 ms, the pipeline's fill and drain were a tenth of the row: 42.8-47.5 / 40.5-46.2 / 40.4-44.6 / 36.3-39.4); `host_fed` {n1['host_fed']['metrics']['GBps_over_pcie']:.1f} GB/s; `boundary_per_message` {bpm['c4']['value']/1e6:.0f} M messages/s ({bpm['c4']['ns_per_message']:.1f} ns
each; config 4's records; other boxes 274-296 M) and {bpm['c3_alive_keys']['value']/1e6:.0f} M/s with `-c` and 16-byte keys ({bpm['c3_alive_keys']['ns_per_message']:.1f} ns) on one host thread.
`r06_bench_c5_forced.json`: {c5f['ms_per_step']:.1f} ms per step on one rank (100 M distinct keys on ONE GPU in the table state — the direct path carries most of
the batch; the exchange re-sends every entry the rank ever wrote); `r06_bench_c4_strong_forced.json`: {c4f['ms_per_step']:.2f} ms per step.
'''
pth=root+'README.md'; s=open(pth).read()
a=s.index("Round 6 (final kernels; `r06_*` from ONE gpurun call")
open(pth,'w').write(s[:a]+txt)
print(txt[:3000])
