// rdkafka_source.cpp — see rdkafka_source.hpp.  librdkafka is bound at run time (dlopen): the
// declarations below restate the public C API of librdkafka 1.x (rdkafka.h) that this file uses.
#include "rdkafka_source.hpp"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace kta {

namespace {

// ---- librdkafka's public ABI, as far as it is used here ---------------------------------------------
typedef struct rd_kafka_s rd_kafka_t;
typedef struct rd_kafka_topic_s rd_kafka_topic_t;
typedef struct rd_kafka_conf_s rd_kafka_conf_t;
typedef struct rd_kafka_topic_partition_list_s rd_kafka_topic_partition_list_t;
typedef int rd_kafka_resp_err_t;                       // C enum; 0 = RD_KAFKA_RESP_ERR_NO_ERROR
enum { RD_KAFKA_CONSUMER = 1 };                        // rd_kafka_type_t
enum { RD_KAFKA_CONF_OK = 0 };                         // rd_kafka_conf_res_t
enum { RD_KAFKA_TIMESTAMP_NOT_AVAILABLE = 0 };         // rd_kafka_timestamp_type_t
enum { RD_KAFKA_PARTITION_UA = -1 };

struct rd_kafka_message_t {                            // rdkafka.h: struct rd_kafka_message_s
    rd_kafka_resp_err_t err;
    rd_kafka_topic_t *rkt;
    int32_t partition;
    void *payload;                                     // NULL == no payload (tombstone)
    size_t len;
    void *key;                                         // NULL == no key
    size_t key_len;
    int64_t offset;
    void *_private;
};
struct rd_kafka_metadata_broker { int32_t id; char *host; int port; };
struct rd_kafka_metadata_partition {
    int32_t id;
    rd_kafka_resp_err_t err;
    int32_t leader;
    int replica_cnt;
    int32_t *replicas;
    int isr_cnt;
    int32_t *isrs;
};
struct rd_kafka_metadata_topic {
    char *topic;
    int partition_cnt;
    rd_kafka_metadata_partition *partitions;
    rd_kafka_resp_err_t err;
};
struct rd_kafka_metadata {
    int broker_cnt;
    rd_kafka_metadata_broker *brokers;
    int topic_cnt;
    rd_kafka_metadata_topic *topics;
    int32_t orig_broker_id;
    char *orig_broker_name;
};

}  // namespace

struct TopicAnalyzer::Api {
    void *lib = nullptr;
    rd_kafka_conf_t *(*conf_new)() = nullptr;
    int (*conf_set)(rd_kafka_conf_t *, const char *, const char *, char *, size_t) = nullptr;
    void (*conf_destroy)(rd_kafka_conf_t *) = nullptr;
    rd_kafka_t *(*rk_new)(int, rd_kafka_conf_t *, char *, size_t) = nullptr;
    void (*set_log_level)(rd_kafka_t *, int) = nullptr;
    rd_kafka_resp_err_t (*poll_set_consumer)(rd_kafka_t *) = nullptr;
    rd_kafka_topic_t *(*topic_new)(rd_kafka_t *, const char *, void *) = nullptr;
    void (*topic_destroy)(rd_kafka_topic_t *) = nullptr;
    rd_kafka_resp_err_t (*metadata)(rd_kafka_t *, int, rd_kafka_topic_t *, const rd_kafka_metadata **, int) = nullptr;
    void (*metadata_destroy)(const rd_kafka_metadata *) = nullptr;
    rd_kafka_resp_err_t (*query_watermark_offsets)(rd_kafka_t *, const char *, int32_t, int64_t *, int64_t *, int) = nullptr;
    rd_kafka_topic_partition_list_t *(*tpl_new)(int) = nullptr;
    void *(*tpl_add)(rd_kafka_topic_partition_list_t *, const char *, int32_t) = nullptr;
    void (*tpl_destroy)(rd_kafka_topic_partition_list_t *) = nullptr;
    rd_kafka_resp_err_t (*subscribe)(rd_kafka_t *, const rd_kafka_topic_partition_list_t *) = nullptr;
    rd_kafka_message_t *(*consumer_poll)(rd_kafka_t *, int) = nullptr;
    int64_t (*message_timestamp)(const rd_kafka_message_t *, int *) = nullptr;
    void (*message_destroy)(rd_kafka_message_t *) = nullptr;
    rd_kafka_resp_err_t (*offset_store)(rd_kafka_topic_t *, int32_t, int64_t) = nullptr;
    const char *(*err2str)(rd_kafka_resp_err_t) = nullptr;
    rd_kafka_resp_err_t (*consumer_close)(rd_kafka_t *) = nullptr;
    void (*destroy)(rd_kafka_t *) = nullptr;
};

namespace {

template <typename F> bool bind(void *lib, const char *name, F *slot, std::string *missing)
{
    *slot = reinterpret_cast<F>(dlsym(lib, name));
    if (!*slot && missing->empty()) *missing = name;
    return *slot != nullptr;
}

// uuid::Uuid::new_v4() (kafka.rs:27): 122 random bits, version 4, RFC 4122 variant, hyphenated lower case
std::string uuid_v4()
{
    uint8_t b[16];
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(b, 1, 16, f) != 16)
        for (int i = 0; i < 16; i++) b[i] = (uint8_t)rand();
    if (f) fclose(f);
    b[6] = (uint8_t)((b[6] & 0x0F) | 0x40);
    b[8] = (uint8_t)((b[8] & 0x3F) | 0x80);
    char out[37];
    snprintf(out, sizeof(out), "%02x%02x%02x%02x-%02x%02x-%02x%02x-%02x%02x-%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2],
             b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
    return out;
}

}  // namespace

TopicAnalyzer *TopicAnalyzer::new_from_bootstrap_servers(const std::string &bootstrap_server,
                                                         const std::map<std::string, std::string> &librdkafka_settings)
{
    const char *loc = "src/kafka.rs:51";
    Api *api = new Api();
    const char *override_name = getenv("KTA_RDKAFKA_LIB");
    const char *names[] = {override_name, "librdkafka.so.1", "librdkafka.so"};
    std::string why;
    for (const char *n : names) {
        if (!n || !*n) continue;
        api->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api->lib) break;
        if (why.empty()) why = dlerror();
    }
    if (!api->lib) {
        delete api;
        throw RustPanic("Consumer creation failed: librdkafka could not be loaded (" + why + ")", loc);
    }
    std::string missing;
    bool ok = true;
    ok &= bind(api->lib, "rd_kafka_conf_new", &api->conf_new, &missing);
    ok &= bind(api->lib, "rd_kafka_conf_set", &api->conf_set, &missing);
    ok &= bind(api->lib, "rd_kafka_conf_destroy", &api->conf_destroy, &missing);
    ok &= bind(api->lib, "rd_kafka_new", &api->rk_new, &missing);
    ok &= bind(api->lib, "rd_kafka_set_log_level", &api->set_log_level, &missing);
    ok &= bind(api->lib, "rd_kafka_poll_set_consumer", &api->poll_set_consumer, &missing);
    ok &= bind(api->lib, "rd_kafka_topic_new", &api->topic_new, &missing);
    ok &= bind(api->lib, "rd_kafka_topic_destroy", &api->topic_destroy, &missing);
    ok &= bind(api->lib, "rd_kafka_metadata", &api->metadata, &missing);
    ok &= bind(api->lib, "rd_kafka_metadata_destroy", &api->metadata_destroy, &missing);
    ok &= bind(api->lib, "rd_kafka_query_watermark_offsets", &api->query_watermark_offsets, &missing);
    ok &= bind(api->lib, "rd_kafka_topic_partition_list_new", &api->tpl_new, &missing);
    ok &= bind(api->lib, "rd_kafka_topic_partition_list_add", &api->tpl_add, &missing);
    ok &= bind(api->lib, "rd_kafka_topic_partition_list_destroy", &api->tpl_destroy, &missing);
    ok &= bind(api->lib, "rd_kafka_subscribe", &api->subscribe, &missing);
    ok &= bind(api->lib, "rd_kafka_consumer_poll", &api->consumer_poll, &missing);
    ok &= bind(api->lib, "rd_kafka_message_timestamp", &api->message_timestamp, &missing);
    ok &= bind(api->lib, "rd_kafka_message_destroy", &api->message_destroy, &missing);
    ok &= bind(api->lib, "rd_kafka_offset_store", &api->offset_store, &missing);
    ok &= bind(api->lib, "rd_kafka_err2str", &api->err2str, &missing);
    ok &= bind(api->lib, "rd_kafka_consumer_close", &api->consumer_close, &missing);
    ok &= bind(api->lib, "rd_kafka_destroy", &api->destroy, &missing);
    if (!ok) {
        dlclose(api->lib);
        delete api;
        throw RustPanic("Consumer creation failed: librdkafka lacks " + missing, loc);
    }

    // kafka.rs:24-43: the reference's fixed settings, then the user's --librdkafka pairs on top
    const char *user = getenv("USER");   // the reference bakes env!("USER") in at compile time
    std::vector<std::pair<std::string, std::string>> cfg = {
        {"group.id", "topic-analyzer--" + std::string(user ? user : "") + "-" + uuid_v4()},
        {"bootstrap.servers", bootstrap_server},
        {"enable.partition.eof", "false"},
        {"auto.offset.reset", "earliest"},
        {"enable.auto.commit", "false"},
        {"api.version.request", "true"},
        {"enable.auto.offset.store", "false"},
        {"client.id", "topic-analyzer"},
        {"queue.buffering.max.ms", "1000"},
    };
    for (const auto &kv : librdkafka_settings)
        if (kv.first.rfind("kta.", 0) != 0) cfg.push_back(kv);
    rd_kafka_conf_t *conf = api->conf_new();
    char errstr[512] = {0};
    for (const auto &kv : cfg) {
        if (api->conf_set(conf, kv.first.c_str(), kv.second.c_str(), errstr, sizeof(errstr)) != RD_KAFKA_CONF_OK) {
            const std::string msg = std::string("Consumer creation failed: ") + errstr;   // ClientConfig::create() -> Err
            api->conf_destroy(conf);
            dlclose(api->lib);
            delete api;
            throw RustPanic(msg, loc);
        }
    }
    rd_kafka_t *rk = api->rk_new(RD_KAFKA_CONSUMER, conf, errstr, sizeof(errstr));   // takes ownership of conf
    if (!rk) {
        const std::string msg = std::string("Consumer creation failed: ") + errstr;
        api->conf_destroy(conf);
        dlclose(api->lib);
        delete api;
        throw RustPanic(msg, loc);
    }
    api->set_log_level(rk, 6);                   // RDKafkaLogLevel::Info (kafka.rs:49)
    (void)api->poll_set_consumer(rk);            // BaseConsumer: one queue for consumer_poll
    TopicAnalyzer *ta = new TopicAnalyzer();
    ta->api_ = api;
    ta->rk_ = rk;
    return ta;
}

TopicAnalyzer::~TopicAnalyzer()
{
    if (api_) {
        if (rk_) {
            (void)api_->consumer_close(static_cast<rd_kafka_t *>(rk_));
            api_->destroy(static_cast<rd_kafka_t *>(rk_));
        }
        if (api_->lib) dlclose(api_->lib);
        delete api_;
    }
}

void TopicAnalyzer::get_topic_offsets(const std::string &topic, std::map<int32_t, int64_t> *start_offsets,
                                      std::map<int32_t, int64_t> *end_offsets)
{
    rd_kafka_t *rk = static_cast<rd_kafka_t *>(rk_);
    // fetch_metadata(Some(topic), 10 s) (kafka.rs:61)
    rd_kafka_topic_t *rkt = api_->topic_new(rk, topic.c_str(), nullptr);
    const rd_kafka_metadata *md = nullptr;
    const rd_kafka_resp_err_t err = rkt ? api_->metadata(rk, 0, rkt, &md, 10000) : -1;
    if (rkt) api_->topic_destroy(rkt);
    if (err != 0 || !md)
        throw RustPanic(std::string("Error fetching metadata: Meta data fetch error: ") + api_->err2str(err), "src/kafka.rs:61");
    if (md->topic_cnt < 1) {
        api_->metadata_destroy(md);
        throw RustPanic("Topic not found!", "src/kafka.rs:62");
    }
    const rd_kafka_metadata_topic &t = md->topics[0];
    for (int i = 0; i < t.partition_cnt; i++) {           // kafka.rs:66-70
        const int32_t id = t.partitions[i].id;
        int64_t low = 0, high = 0;
        const rd_kafka_resp_err_t werr = api_->query_watermark_offsets(rk, topic.c_str(), id, &low, &high, 1000);
        if (werr != 0) {
            const std::string e = api_->err2str(werr);
            api_->metadata_destroy(md);
            throw RustPanic("called `Result::unwrap()` on an `Err` value: MetadataFetch(" + e + ")", "src/kafka.rs:67");
        }
        (*start_offsets)[id] = low;
        (*end_offsets)[id] = high;
    }
    api_->metadata_destroy(md);
}

uint64_t TopicAnalyzer::read_topic_into_metrics(const std::string &topic, const std::map<int32_t, int64_t> &end_offsets)
{
    rd_kafka_t *rk = static_cast<rd_kafka_t *>(rk_);
    uint64_t seq = 0;
    std::map<int32_t, bool> still_running;                 // kafka.rs:78-82
    for (const auto &kv : end_offsets) still_running[kv.first] = true;

    printf("Subscribing to %s\n", topic.c_str());          // kafka.rs:88
    rd_kafka_topic_partition_list_t *tpl = api_->tpl_new(1);
    (void)api_->tpl_add(tpl, topic.c_str(), RD_KAFKA_PARTITION_UA);
    const rd_kafka_resp_err_t serr = api_->subscribe(rk, tpl);
    api_->tpl_destroy(tpl);
    if (serr != 0)
        throw RustPanic(std::string("Can't subscribe to specified topic: Subscription error: ") + api_->err2str(serr),
                        "src/kafka.rs:89");
    printf("Starting message consumption...\n");           // kafka.rs:91
    fflush(stdout);
    while (true) {
        rd_kafka_message_t *m = api_->consumer_poll(rk, 100);   // kafka.rs:93
        if (!m) continue;                                        // None
        if (m->err != 0) {                                       // Some(Err(e))   kafka.rs:95-97
            fprintf(stderr, "[WARN] Kafka error: Message consumption error: %s\n", api_->err2str(m->err));
            api_->message_destroy(m);
            continue;
        }
        seq += 1;                                                // kafka.rs:99-101
        Message msg;
        msg.partition = m->partition;
        msg.offset = m->offset;
        int tstype = RD_KAFKA_TIMESTAMP_NOT_AVAILABLE;
        const int64_t ts = api_->message_timestamp(m, &tstype);
        msg.timestamp_ms = tstype == RD_KAFKA_TIMESTAMP_NOT_AVAILABLE ? -1 : ts;   // to_millis(): n.a. / -1 -> None -> 0
        msg.key = static_cast<const uint8_t *>(m->key);         // key(): None iff the pointer is null
        msg.key_len = m->key ? (int64_t)m->key_len : -1;
        msg.payload_len = m->payload ? (int64_t)m->len : -1;     // payload(): likewise; bytes are never read
        // kafka.rs:103-105: the DateTime of the progress line.  NaiveDateTime::from_timestamp(timestamp / 1000, 0)
        // [3P chrono 0.4.19] expects a date within the years [-262144, 262143] and panics otherwise — here,
        // before any handler has seen the record.
        const int64_t secs = (msg.timestamp_ms == -1 ? 0 : msg.timestamp_ms) / 1000;
        if (secs < KTA_CHRONO_MIN_SEC || secs > KTA_CHRONO_MAX_SEC) {
            api_->message_destroy(m);
            throw RustPanic("invalid or out-of-range datetime",
                            "chrono-0.4.19/src/naive/datetime.rs (NaiveDateTime::from_timestamp, src/kafka.rs:104)");
        }
        for (MetricHandler *mh : metric_handlers_) mh->handle_message(msg);    // kafka.rs:107-109
        const rd_kafka_resp_err_t oerr = api_->offset_store(m->rkt, m->partition, m->offset);   // kafka.rs:115
        if (oerr != 0) fprintf(stderr, "[WARN] Error while storing offset: %s\n", api_->err2str(oerr));
        const int32_t partition = m->partition;
        const int64_t offset = m->offset;
        api_->message_destroy(m);
        const auto end = end_offsets.find(partition);            // kafka.rs:119: .get(&partition).unwrap()
        if (end == end_offsets.end())
            throw RustPanic("called `Option::unwrap()` on a `None` value", "src/kafka.rs:119");
        if (offset + 1 >= end->second) still_running[partition] = false;
        bool all_done = true;                                    // kafka.rs:123-132
        for (const auto &kv : still_running)
            if (kv.second) all_done = false;
        if (all_done) break;
    }
    fprintf(stderr, "done\n");                                   // kafka.rs:136 (spinner's final message)
    return seq;
}

}  // namespace kta
