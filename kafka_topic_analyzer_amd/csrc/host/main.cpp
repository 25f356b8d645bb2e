// main.cpp — `kta-analyzer`: the reference's command line (src/main.rs:29-180) in front of the
// MI355X metric-accumulation path.  Flags, defaults and the printed report are the reference's:
//
//   -t, --topic <TOPIC>                         (required)           main.rs:37-44
//   -b, --bootstrap-server <BOOTSTRAP_SERVER>   (required)           main.rs:45-52
//       --librdkafka <LIBRDKAFKA>               k=v,k=v              main.rs:53-59, 84-92
//   -c, --count-alive-keys                                           main.rs:60-66, 77-80
//   -h/--help, -V/--version ("0.4.1", main.rs:35)
//
// The record source is chosen by the scheme of --bootstrap-server:
//     <host:port,...>                    a Kafka cluster, through the reference's consume loop (src/kafka.rs)
//                                        over librdkafka, which is bound at run time (host/rdkafka_source.hpp);
//                                        "Consumer creation failed" when the library is not installed
//     synthetic://<c1..c5>[?records=N]   the synthetic topic of include/kta_synth.h
//     dump://<path>                      a KTADUMP1 topic dump (host/dump.hpp)
//     segment://<f0>[,<f1>...]           raw Kafka log segments (`*.log` files of a broker, record-batch
//                                        v2; uncompressed, gzip, Snappy, LZ4, zstd): file k is partition k; decoded ON THE GPU
//                                        (include/kta_kafka.h), the host only walks batch headers
// Extra knobs travel in --librdkafka as kta.* keys (kta.device=N, kta.gpus=N,
// kta.batch=N, kta.write_dump=<path>, kta.per_message=1), so no flag is added or renamed.
// kta.gpus=N (synthetic:// and segment:// sources) shards the topic's partitions over N GPUs, partition p on
// rank p % N, one host thread + one context + one communicator rank per GPU (device (kta.device + r) mod the
// visible devices), and replaces "the report reads the handlers" by ONE exchange step (kta_exchange: RCCL).
// kta.per_message=1 drives the handler exactly like the reference's loop (kafka.rs:107-109): one
// MetricHandler::handle_message call per record instead of filling columns.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "dump.hpp"
#include "kta_kafka.h"
#include "kta_synth.h"
#include "metric.hpp"
#include "rdkafka_source.hpp"

namespace {

const char *kAbout = "Kafka Topic Analyzer 0.4.1";

void print_help()
{
    printf("%s\n\nUSAGE:\n    kafka-topic-analyzer [FLAGS] [OPTIONS] --bootstrap-server <BOOTSTRAP_SERVER> --topic <TOPIC>\n\n"
           "FLAGS:\n"
           "    -c, --count-alive-keys    Counts the effective number of alive keys in a log compacted topic by saving the "
           "state for each key in a local file and counting the result at the end of the read operation.A key is 'alive' "
           "when it is present and has a non-null value in it's latest-offset version\n"
           "    -h, --help                Prints help information\n"
           "    -V, --version             Prints version information\n\n"
           "OPTIONS:\n"
           "    -b, --bootstrap-server <BOOTSTRAP_SERVER>    Bootstrap server(s) to work with, comma separated\n"
           "        --librdkafka <LIBRDKAFKA>                Options to pass into the underlying librdkafka, comma "
           "seperated key=value pairs\n"
           "    -t, --topic <TOPIC>                          The topic to analyze\n",
           kAbout);
}

[[noreturn]] void usage_error(const std::string &msg)
{
    fprintf(stderr, "error: %s\n\nUSAGE:\n    kafka-topic-analyzer [FLAGS] [OPTIONS] --bootstrap-server "
                    "<BOOTSTRAP_SERVER> --topic <TOPIC>\n\nFor more information try --help\n", msg.c_str());
    exit(1);
}

std::mutex g_rank_panic_mutex;           // a panic of a rank thread of a sharded run, kept for the main thread
bool g_rank_panicked = false;
std::string g_rank_panic_msg, g_rank_panic_loc;

[[noreturn]] void rust_panic(const std::string &msg, const std::string &loc)
{
    fprintf(stderr, "thread 'main' panicked at '%s', %s\n", msg.c_str(), loc.c_str());
    exit(101);
}

struct Args {
    std::string topic, bootstrap, librdkafka;
    bool has_topic = false, has_bootstrap = false, has_librdkafka = false;
    int count_alive_occurrences = 0;
};

Args parse_args(int argc, char **argv)
{
    Args a;
    auto take = [&](int &i, const std::string &name, const char *inline_val) -> std::string {
        if (inline_val) return inline_val;
        if (i + 1 >= argc) usage_error("The argument '" + name + "' requires a value but none was supplied");
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        const char *eq = nullptr;
        std::string name = s;
        if (s.rfind("--", 0) == 0) {
            size_t p = s.find('=');
            if (p != std::string::npos) { name = s.substr(0, p); eq = argv[i] + p + 1; }
        } else if (s.size() > 2 && s[0] == '-' && s[1] != '-') {
            name = s.substr(0, 2);
            eq = argv[i] + 2;
            if (name == "-c" || name == "-h" || name == "-V") { eq = nullptr; name = s; }
        }
        if (name == "-t" || name == "--topic") { a.topic = take(i, "--topic <TOPIC>", eq); a.has_topic = true; }
        else if (name == "-b" || name == "--bootstrap-server") { a.bootstrap = take(i, "--bootstrap-server <BOOTSTRAP_SERVER>", eq); a.has_bootstrap = true; }
        else if (name == "--librdkafka") { a.librdkafka = take(i, "--librdkafka <LIBRDKAFKA>", eq); a.has_librdkafka = true; }
        else if (name == "-c" || name == "--count-alive-keys") {
            if (++a.count_alive_occurrences > 1)   // clap 2: a flag that is not `multiple(true)` may appear once
                usage_error("The argument '--count-alive-keys' was provided more than once, but cannot be used multiple times");
        }
        else if (name == "-h" || name == "--help") { print_help(); exit(0); }
        else if (name == "-V" || name == "--version") { printf("%s\n", kAbout); exit(0); }
        else usage_error("Found argument '" + s + "' which wasn't expected, or isn't valid in this context");
    }
    if (!a.has_topic || !a.has_bootstrap) {
        std::string m = "The following required arguments were not provided:";
        if (!a.has_bootstrap) m += "\n    --bootstrap-server <BOOTSTRAP_SERVER>";
        if (!a.has_topic) m += "\n    --topic <TOPIC>";
        usage_error(m);
    }
    return a;
}

// main.rs:84-92: split(",") then split('=') taking the first two pieces; a pair without '=' panics
std::map<std::string, std::string> parse_librdkafka(const Args &a)
{
    std::map<std::string, std::string> m;
    if (!a.has_librdkafka) return m;
    size_t pos = 0;
    const std::string &s = a.librdkafka;
    while (true) {
        size_t comma = s.find(',', pos);
        std::string kv = s.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        size_t eq = kv.find('=');
        if (eq == std::string::npos)
            rust_panic("called `Option::unwrap()` on a `None` value", "src/main.rs:89:48");
        size_t eq2 = kv.find('=', eq + 1);
        m[kv.substr(0, eq)] = kv.substr(eq + 1, eq2 == std::string::npos ? std::string::npos : eq2 - eq - 1);
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return m;
}

void check(int rc, kta_ctx *ctx, const char *what)
{
    if (rc != KTA_OK) {
        fprintf(stderr, "%s failed: %s\n", what, kta_last_error(ctx));
        exit(2);
    }
}

// ---- kta.gpus=N: one rank per GPU ------------------------------------------------------------------------
struct ShardedJob {
    std::string topic;
    bool count_alive = false, synthetic = false, check_crcs = false;
    int device = 0, nranks = 1;
    uint64_t batch = 1ull << 20;
    uint32_t P = 0;
    kta_synth_spec spec{};
    bool oversubscribe = false;                        // kta.oversubscribe=1: several ranks may share a device (test doubles of RCCL)
    uint64_t n_records = 0;
    std::vector<std::vector<uint8_t>> segment_bytes;   // segment:// : file k is partition k
    std::vector<uint64_t> base_seq;                    //   global sequence number of each partition's first record
    std::vector<int64_t> start_offsets, end_offsets;
};

// What one rank consumes: the partitions p with p % nranks == rank, every record with its GLOBAL sequence
// number (the position the single-GPU run consumes it at), then the exchange.
void run_rank(const ShardedJob &job, int rank, const uint8_t *uid, int ndev, kta::HipMetricHandler **out)
{
    try {
        // a rank's records are not consecutive in consumption order: global sequence numbers, table state
        const uint32_t flags = job.count_alive ? (job.synthetic ? KTA_FLAG_SEQ_COLUMN : KTA_FLAG_ALIVE_TABLE) : 0u;
        kta::HipMetricHandler *h = new kta::HipMetricHandler((int32_t)job.P, job.count_alive, (job.device + rank) % ndev,
                                                             job.batch, 0, flags);
        kta_ctx *ctx = h->ctx();
        h->comm_create(job.nranks, rank, uid);
        if (job.synthetic) {
            // every rank enumerates the whole topic in consumption order and keeps its partitions
            std::vector<int32_t> part(job.batch), kl(job.batch), vl(job.batch);
            std::vector<int64_t> ts(job.batch);
            std::vector<uint32_t> ko(job.batch);
            std::vector<uint8_t> kbuf;
            kta_batch hb{};
            bool open = false;
            uint64_t fill = 0, fill_kb = 0;
            auto submit = [&]() {
                if (open) check(kta_batch_submit(ctx, fill, fill_kb, 0), ctx, "kta_batch_submit");
                open = false;
                fill = fill_kb = 0;
            };
            for (uint64_t at = 0; at < job.n_records;) {
                uint64_t n = std::min<uint64_t>(job.batch, job.n_records - at), kb = 0;
                kta_batch tmp{};
                tmp.partition = part.data(); tmp.key_len = kl.data(); tmp.val_len = vl.data(); tmp.ts_ms = ts.data();
                tmp.capacity = n;
                if (job.count_alive) {
                    check(kta_synth_fill_host(&job.spec, at, n, &tmp, &kb), ctx, "kta_synth_fill_host");   // key bytes needed
                    kbuf.resize(kb + 16);
                    tmp.key_off = ko.data(); tmp.key_bytes = kbuf.data(); tmp.key_bytes_capacity = kb;
                }
                check(kta_synth_fill_host(&job.spec, at, n, &tmp, &kb), ctx, "kta_synth_fill_host");
                for (uint64_t i = 0; i < n; i++) {
                    if (part[i] % job.nranks != rank) continue;
                    const uint64_t klen = job.count_alive && kl[i] > 0 ? (uint64_t)kl[i] : 0;
                    if (open && (fill == hb.capacity || fill_kb + klen > hb.key_bytes_capacity)) submit();
                    if (!open) {
                        check(kta_batch_acquire(ctx, &hb), ctx, "kta_batch_acquire");
                        open = true;
                    }
                    hb.partition[fill] = part[i]; hb.key_len[fill] = kl[i]; hb.val_len[fill] = vl[i]; hb.ts_ms[fill] = ts[i];
                    if (job.count_alive) {
                        hb.key_off[fill] = (uint32_t)fill_kb;
                        if (klen) memcpy(hb.key_bytes + fill_kb, kbuf.data() + ko[i], klen);
                        fill_kb += klen;
                        hb.seq[fill] = at + i;
                    }
                    fill++;
                }
                at += n;
            }
            submit();
        } else {
            check(kta_kafka_set_check_crcs(ctx, job.check_crcs ? 1 : 0), ctx, "kta_kafka_set_check_crcs");
            for (uint32_t p = (uint32_t)rank; p < job.P; p += (uint32_t)job.nranks) {
                if (job.segment_bytes[p].empty()) continue;
                kta_kafka_index_stats ist;
                check(kta_seek_seq(ctx, job.base_seq[p]), ctx, "kta_seek_seq");
                check(kta_kafka_consume(ctx, job.segment_bytes[p].data(), job.segment_bytes[p].size(), (int32_t)p, &ist), ctx,
                      "kta_kafka_consume");
                if (ist.n_compressed || ist.n_old_magic)
                    fprintf(stderr, "[WARN] Kafka error: partition %u: %llu unknown-codec and %llu pre-v2 batches skipped\n", p,
                            (unsigned long long)ist.n_compressed, (unsigned long long)ist.n_old_magic);
            }
        }
        h->exchange(!job.synthetic);
        *out = h;
    } catch (const kta::RustPanic &p) {
        // Every rank sees the job's extrema after the exchange, so all of them end here — and exit() is not to be
        // called from several threads at once (handlers run at exit, the HIP runtime is torn down).  The panic is
        // kept for the main thread, which prints it once after joining the ranks, like the reference's one line.
        std::lock_guard<std::mutex> lock(g_rank_panic_mutex);
        if (!g_rank_panicked) {
            g_rank_panicked = true;
            g_rank_panic_msg = p.what();
            g_rank_panic_loc = p.location;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "rank %d: %s\n", rank, e.what());
        exit(2);   // before the exchange: the other ranks would wait for this one for ever
    }
}

int run_sharded(ShardedJob &job, const std::chrono::steady_clock::time_point start_time)
{
    int ndev = 0;
    if (kta_device_count(&ndev) != KTA_OK) {
        fprintf(stderr, "kta_create failed: no HIP device visible (libkta_hip has no CPU fallback)\n");
        return 2;
    }
    // one rank per device: two ranks on one GPU would each take a 32 GiB table under -c, and real RCCL refuses
    // the second one ("duplicate GPU") while the first is still blocked in its init
    if (job.nranks > ndev && !job.oversubscribe) {
        fprintf(stderr, "kta.gpus=%d: only %d HIP device(s) visible\n", job.nranks, ndev);
        return 2;
    }
    uint8_t uid[KTA_COMM_ID_BYTES];
    if (kta_comm_unique_id(uid) != KTA_OK) {
        fprintf(stderr, "kta.gpus=%d: RCCL is not loadable (KTA_RCCL_LIBRARY)\n", job.nranks);
        return 2;
    }
    printf("Subscribing to %s\n", job.topic.c_str());                              // kafka.rs:88
    printf("Starting message consumption...\n");                                   // kafka.rs:91
    fflush(stdout);
    std::vector<kta::HipMetricHandler *> handlers(job.nranks, nullptr);
    std::vector<std::thread> threads;
    for (int r = 0; r < job.nranks; r++) threads.emplace_back(run_rank, std::cref(job), r, uid, ndev, &handlers[r]);
    for (auto &t : threads) t.join();
    if (g_rank_panicked) rust_panic(g_rank_panic_msg, g_rank_panic_loc);           // once, from the main thread
    fprintf(stderr, "done\n");                                                     // kafka.rs:136 (spinner)
    kta::HipMetricHandler *h0 = handlers[0];            // after the exchange every rank holds the whole job's result
    uint64_t undelivered = 0;
    for (auto *h : handlers) undelivered = std::max(undelivered, h->undelivered_records());
    if (!job.synthetic && undelivered)
        fprintf(stderr, "[WARN] Kafka error: %llu record(s) of corrupt batches were not delivered\n", (unsigned long long)undelivered);
    const kta::MessageMetrics &metrics = h0->metrics();
    if (job.synthetic)
        for (uint32_t p = 0; p < job.P; p++) job.end_offsets[p] = (int64_t)metrics.total((int32_t)p);
    std::vector<int32_t> partitions(job.P);
    for (uint32_t p = 0; p < job.P; p++) partitions[p] = (int32_t)p;
    const uint64_t duration_secs = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(
                                       std::chrono::steady_clock::now() - start_time).count();
    try {
        std::string text = kta::render_report(job.topic, duration_secs, metrics, h0->log_compaction(), partitions,
                                              job.start_offsets, job.end_offsets);
        fputs(text.c_str(), stdout);
    } catch (const kta::RustPanic &p) {
        rust_panic(p.what(), p.location);
    }
    for (auto *h : handlers) delete h;
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    Args args = parse_args(argc, argv);
    const auto start_time = std::chrono::steady_clock::now();                       // main.rs:69
    const bool count_alive = args.count_alive_occurrences == 1;                     // main.rs:77-80
    std::map<std::string, std::string> cfg = parse_librdkafka(args);                // main.rs:84-92
    const int device = cfg.count("kta.device") ? atoi(cfg["kta.device"].c_str()) : 0;
    const uint64_t batch = cfg.count("kta.batch") ? strtoull(cfg["kta.batch"].c_str(), nullptr, 10) : (1ull << 20);
    int gpus = 1;
    if (cfg.count("kta.gpus")) {   // strictly a positive decimal number
        const std::string &g = cfg["kta.gpus"];
        char *end = nullptr;
        const long v = strtol(g.c_str(), &end, 10);
        if (g.empty() || *end || v < 1 || v > 1024) {
            fprintf(stderr, "kta.gpus=%s: expected a number of GPUs >= 1\n", g.c_str());
            return 2;
        }
        gpus = (int)v;
    }
    const bool oversubscribe = cfg.count("kta.oversubscribe") && cfg["kta.oversubscribe"] == "1";

    // ---- the record source (stands in for TopicAnalyzer, src/kafka.rs) ------------------------------
    const std::string &b = args.bootstrap;
    bool synthetic = b.rfind("synthetic://", 0) == 0, dump = b.rfind("dump://", 0) == 0;
    const bool segment = b.rfind("segment://", 0) == 0;
    const bool kafka = !synthetic && !dump && !segment;   // a broker list: the reference's own path
    kta::TopicAnalyzer *topic_analyzer = nullptr;
    std::map<int32_t, int64_t> kafka_start, kafka_end;
    if (kafka) {
        try {
            topic_analyzer = kta::TopicAnalyzer::new_from_bootstrap_servers(b, cfg);      // main.rs:93
            topic_analyzer->get_topic_offsets(args.topic, &kafka_start, &kafka_end);      // main.rs:94-96
        } catch (const kta::RustPanic &p) {
            rust_panic(p.what(), p.location);
        }
    }
    std::vector<std::string> segment_files;
    if (segment) {
        std::string rest = b.substr(strlen("segment://"));
        size_t p0 = 0;
        while (true) {
            size_t c = rest.find(',', p0);
            segment_files.push_back(rest.substr(p0, c == std::string::npos ? std::string::npos : c - p0));
            if (c == std::string::npos) break;
            p0 = c + 1;
        }
    }
    kta_synth_spec spec{};
    uint64_t n_records = 0;
    kta::DumpHeader hdr;
    kta::DumpReader *reader = nullptr;
    if (synthetic) {
        std::string rest = b.substr(strlen("synthetic://"));
        std::string preset = rest.substr(0, rest.find('?'));
        if (kta_synth_preset(preset.c_str(), &spec, &n_records) != KTA_OK) {
            fprintf(stderr, "Error fetching metadata: unknown synthetic topic '%s'\n", preset.c_str());
            return 101;
        }
        size_t q = rest.find("records=");
        if (q != std::string::npos) n_records = strtoull(rest.c_str() + q + 8, nullptr, 10);
        hdr.n_partitions = spec.n_partitions;
        hdr.n_records = n_records;
    } else if (segment) {
        hdr.n_partitions = (uint32_t)segment_files.size();
        n_records = 1;  // unknown until decoded; non-zero so the emptiness test below looks at the files
    } else if (kafka) {
        hdr.n_partitions = kafka_end.empty() ? 0u : (uint32_t)(kafka_end.rbegin()->first + 1);   // ids are dense
    } else {
        reader = new kta::DumpReader(b.substr(strlen("dump://")));
        if (!reader->ok() || !reader->read_header(&hdr)) {
            fprintf(stderr, "Error fetching metadata: cannot read topic dump '%s'\n", b.c_str() + 7);
            return 101;
        }
    }
    const uint32_t P = hdr.n_partitions;

    // librdkafka options are forwarded as in the reference (kafka.rs:38-42); the one this build can
    // honour for raw segments is check.crcs (default false): verify every batch's CRC-32C on the GPU
    const bool check_crcs = cfg.count("check.crcs") && cfg["check.crcs"] == "true";
    if (gpus > 1 && !synthetic && !segment) {
        fprintf(stderr, "kta.gpus=%d needs a synthetic:// or segment:// source\n", gpus);
        return 2;
    }
    std::vector<int64_t> start_offsets(P, 0), end_offsets(P, 0);
    if (dump) { start_offsets = hdr.start_offsets; end_offsets = hdr.end_offsets; }
    if (kafka)
        for (const auto &kv : kafka_end) {
            start_offsets[(size_t)kv.first] = kafka_start[kv.first];
            end_offsets[(size_t)kv.first] = kv.second;
        }
    std::vector<std::vector<uint8_t>> segment_bytes;
    std::vector<uint64_t> segment_base_seq;   // records of the partitions before each one (consumption order: file by file)
    uint64_t segment_records = 0;
    if (segment) {  // watermarks = first / last offset found in each partition's segment (kafka.rs:60-72)
        for (uint32_t p = 0; p < P; p++) {
            std::vector<uint8_t> bytes;
            FILE *f = fopen(segment_files[p].c_str(), "rb");
            if (!f) {
                fprintf(stderr, "Error fetching metadata: cannot read log segment '%s'\n", segment_files[p].c_str());
                return 101;
            }
            fseek(f, 0, SEEK_END);
            long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            bytes.resize((size_t)sz + 64, 0);
            if (sz > 0 && fread(bytes.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 101; }
            fclose(f);
            bytes.resize((size_t)sz);
            kta_kafka_index_stats ist;
            std::vector<kta_kafka_batch_desc> descs(1);
            int rc = kta_kafka_index_host(bytes.data(), bytes.size(), (int32_t)p, 0, 0, 0, descs.data(), descs.size(), &ist);
            if (rc == KTA_ERR_CAPACITY) {
                descs.resize(ist.n_batches);
                rc = kta_kafka_index_host(bytes.data(), bytes.size(), (int32_t)p, 0, 0, 0, descs.data(), descs.size(), &ist);
            }
            if (rc == KTA_OK && ist.any_offsets) {   // what the broker would answer: log start / log end offset,
                start_offsets[p] = ist.first_offset;  // control batches and compacted-away offsets included
                end_offsets[p] = ist.next_offset;
            }
            segment_base_seq.push_back(segment_records);
            if (rc == KTA_OK) segment_records += ist.n_records;
            segment_bytes.push_back(std::move(bytes));
        }
    }
    else if (synthetic && n_records > 0) {
        // offsets of a synthetic topic: 0 .. per-partition record count; known only after the scan,
        // so the emptiness test of main.rs:98-101 uses the record count
        std::fill(end_offsets.begin(), end_offsets.end(), 1);
    }
    if (std::all_of(end_offsets.begin(), end_offsets.end(), [](int64_t v) { return v == 0; })) {  // main.rs:98-101
        fprintf(stderr, "[ERROR] Given topic has no content, no analysis possible. Exiting.\n");
        return 254;  // exit(-2)
    }
    std::vector<int32_t> partitions(P);                                             // main.rs:103-106
    for (uint32_t p = 0; p < P; p++) partitions[p] = (int32_t)p;

    if (gpus > 1) {   // kta.gpus=N: partition p on rank p % N, one exchange step before the report
        ShardedJob job;
        job.topic = args.topic;
        job.count_alive = count_alive;
        job.synthetic = synthetic;
        job.check_crcs = check_crcs;
        job.device = device;
        job.nranks = gpus;
        job.oversubscribe = oversubscribe;
        job.batch = batch;
        job.P = P;
        job.spec = spec;
        job.n_records = n_records;
        job.segment_bytes = std::move(segment_bytes);
        job.base_seq = segment_base_seq;
        job.start_offsets = start_offsets;
        job.end_offsets = end_offsets;
        return run_sharded(job, start_time);
    }
    // the handlers are created only now: MessageMetrics::new / LogCompactionInMemoryMetrics::new come after the
    // "no content" exit in the reference as well (main.rs:77-82 build them, but nothing is allocated there; here
    // -c means a 32 GiB table)
    kta::HipMetricHandler *handler = nullptr;
    try {
        handler = new kta::HipMetricHandler((int32_t)P, count_alive, device, batch, 0);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 2;
    }
    kta_ctx *ctx = handler->ctx();
    if (segment) check(kta_kafka_set_check_crcs(ctx, check_crcs ? 1 : 0), ctx, "kta_kafka_set_check_crcs");

    if (!kafka) {
        printf("Subscribing to %s\n", args.topic.c_str());                          // kafka.rs:88
        printf("Starting message consumption...\n");                                // kafka.rs:91
        fflush(stdout);
    }

    kta::DumpWriter *writer = nullptr;
    std::vector<kta::DumpBatch> to_write;
    const bool write_dump = cfg.count("kta.write_dump") != 0;

    uint64_t seq = 0;
    const bool per_message = cfg.count("kta.per_message") && cfg["kta.per_message"] == "1";
    if (kafka) {
        // the reference's loop: every polled message goes to every handler (kafka.rs:92-135); the handler
        // stages it into pinned columns and the device accumulates batch by batch
        topic_analyzer->add_metric_handler(handler);                               // main.rs:108-115
        try {
            seq = topic_analyzer->read_topic_into_metrics(args.topic, kafka_end);  // main.rs:117
        } catch (const kta::RustPanic &p) {
            rust_panic(p.what(), p.location);
        } catch (const std::exception &e) {
            fprintf(stderr, "%s\n", e.what());
            return 2;
        }
        delete topic_analyzer;                                                     // end of the block, main.rs:118
        topic_analyzer = nullptr;
    } else if (synthetic && per_message) {
        // the reference's shape: for every polled message, every handler's handle_message (kafka.rs:107-109)
        std::vector<kta::MetricHandler *> metric_handlers{handler};
        std::vector<uint8_t> key;
        for (; seq < n_records; seq++) {
            kta::Message m;
            int32_t p, kl, vl;
            int64_t ts;
            kta_synth_record(&spec, seq, &p, &kl, &vl, &ts);
            m.partition = p;
            m.offset = (int64_t)seq;
            m.timestamp_ms = ts;
            m.payload_len = vl;
            if (kl >= 0) {
                key.resize((size_t)kl + 1);
                const uint64_t kid = (uint64_t)kta_synth_key_id(&spec, seq);
                for (int32_t j = 0; j < kl; j++) key[(size_t)j] = kta_synth_key_byte(&spec, kid, (uint32_t)j);
                m.key = key.data();   // non-null even for an empty key: Some(&[])
                m.key_len = kl;
            }
            try {
                for (auto *mh : metric_handlers) mh->handle_message(m);
            } catch (const kta::RustPanic &p) {
                rust_panic(p.what(), p.location);
            } catch (const std::exception &e) {
                fprintf(stderr, "%s\n", e.what());
                return 2;
            }
        }
    } else if (synthetic) {
        while (seq < n_records) {
            kta_batch hb;
            check(kta_batch_acquire(ctx, &hb), ctx, "kta_batch_acquire");
            uint64_t n = std::min<uint64_t>(hb.capacity, n_records - seq), kb = 0;
            if (!count_alive && !write_dump) { hb.key_off = nullptr; hb.key_bytes = nullptr; }
            int rc = kta_synth_fill_host(&spec, seq, n, &hb, &kb);
            while (rc == KTA_ERR_CAPACITY && n > 1) {  // key bytes did not fit: shrink the batch
                n /= 2;
                rc = kta_synth_fill_host(&spec, seq, n, &hb, &kb);
            }
            check(rc, ctx, "kta_synth_fill_host");
            if (write_dump) {
                kta::DumpBatch db;
                db.n = n; db.n_key_bytes = hb.key_bytes ? kb : 0;
                db.partition.assign(hb.partition, hb.partition + n);
                db.key_len.assign(hb.key_len, hb.key_len + n);
                db.val_len.assign(hb.val_len, hb.val_len + n);
                db.ts_ms.assign(hb.ts_ms, hb.ts_ms + n);
                if (hb.key_off) db.key_off.assign(hb.key_off, hb.key_off + n); else db.key_off.assign(n, 0);
                if (hb.key_bytes) db.key_bytes.assign(hb.key_bytes, hb.key_bytes + kb);
                to_write.push_back(std::move(db));
            }
            check(kta_batch_submit(ctx, n, kb, seq), ctx, "kta_batch_submit");
            seq += n;
        }
    } else if (segment) {
        for (uint32_t p = 0; p < P; p++) {
            if (segment_bytes[p].empty()) continue;
            kta_kafka_index_stats ist;
            check(kta_kafka_consume(ctx, segment_bytes[p].data(), segment_bytes[p].size(), (int32_t)p, &ist), ctx,
                  "kta_kafka_consume");
            if (ist.n_compressed || ist.n_old_magic)
                fprintf(stderr, "[WARN] Kafka error: partition %u: %llu unknown-codec and %llu pre-v2 batches skipped\n", p,
                        (unsigned long long)ist.n_compressed, (unsigned long long)ist.n_old_magic);   // kafka.rs:95-97
            seq += ist.n_records;
        }
    } else {
        kta::DumpBatch db;
        for (uint64_t bi = 0; bi < hdr.n_batches; bi++) {
            if (!reader->read_batch(&db)) {
                fprintf(stderr, "[WARN] Kafka error: truncated topic dump\n");      // kafka.rs:95-97: warn and go on
                break;
            }
            uint64_t done = 0;
            while (done < db.n) {
                kta_batch hb;
                check(kta_batch_acquire(ctx, &hb), ctx, "kta_batch_acquire");
                uint64_t n = std::min<uint64_t>(hb.capacity, db.n - done), kb = 0;
                if (count_alive) {  // re-pack this chunk's keys
                    uint64_t m = 0;
                    for (; m < n; m++) {
                        const uint64_t kl = db.key_len[done + m] > 0 ? (uint64_t)db.key_len[done + m] : 0;
                        if (kb + kl > hb.key_bytes_capacity) break;
                        hb.key_off[m] = (uint32_t)kb;
                        if (kl) memcpy(hb.key_bytes + kb, db.key_bytes.data() + db.key_off[done + m], kl);
                        kb += kl;
                    }
                    n = m;
                    if (n == 0) { fprintf(stderr, "key larger than the staging capacity\n"); return 2; }
                }
                memcpy(hb.partition, db.partition.data() + done, 4 * n);
                memcpy(hb.key_len, db.key_len.data() + done, 4 * n);
                memcpy(hb.val_len, db.val_len.data() + done, 4 * n);
                memcpy(hb.ts_ms, db.ts_ms.data() + done, 8 * n);
                check(kta_batch_submit(ctx, n, kb, seq), ctx, "kta_batch_submit");
                seq += n;
                done += n;
            }
        }
    }
    if (!kafka) fprintf(stderr, "done\n");                                          // kafka.rs:136 (spinner)

    try {
        handler->finish(segment);                // where main.rs:121 is: the trait has no end-of-stream hook
    } catch (const kta::RustPanic &p) {          // a timestamp outside chrono's range: the reference died on that record
        rust_panic(p.what(), p.location);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 2;
    }
    if (segment && handler->undelivered_records()) {                               // kafka.rs:95-97: warn, go on
        uint64_t crc_errors = 0;
        (void)kta_kafka_crc_errors(ctx, &crc_errors);
        fprintf(stderr, "[WARN] Kafka error: %llu record(s) of corrupt batches were not delivered (%llu CRC failure(s))\n",
                (unsigned long long)handler->undelivered_records(), (unsigned long long)crc_errors);
    }
    const kta::MessageMetrics &metrics = handler->metrics();
    if (synthetic)
        for (uint32_t p = 0; p < P; p++) end_offsets[p] = (int64_t)metrics.total((int32_t)p);
    if (write_dump) {
        writer = new kta::DumpWriter(cfg["kta.write_dump"]);
        kta::DumpHeader wh = hdr;
        wh.n_batches = to_write.size();
        wh.start_offsets = start_offsets;
        wh.end_offsets = end_offsets;
        bool ok = writer->ok() && writer->write_header(wh);
        for (auto &db : to_write)
            ok = ok && writer->write_batch(db.n, db.n_key_bytes, db.partition.data(), db.key_len.data(),
                                           db.val_len.data(), db.ts_ms.data(), db.key_off.data(), db.key_bytes.data());
        delete writer;
        if (!ok) fprintf(stderr, "[WARN] could not write topic dump %s\n", cfg["kta.write_dump"].c_str());
    }

    const uint64_t duration_secs = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(
                                       std::chrono::steady_clock::now() - start_time).count();  // main.rs:121
    try {
        std::string text = kta::render_report(args.topic, duration_secs, metrics, handler->log_compaction(),
                                              partitions, start_offsets, end_offsets);
        fputs(text.c_str(), stdout);
    } catch (const kta::RustPanic &p) {
        rust_panic(p.what(), p.location);
    }
    delete handler;
    delete reader;
    return 0;
}
