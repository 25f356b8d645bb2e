// metric.hpp — C++ host mirror of the reference's handler interface over the C ABI (kta_hip.h).
//
//   reference (under /root/reference)                          here
//   src/kafka.rs:18-20   trait MetricHandler                   kta::MetricHandler
//   src/metric.rs:12-26  struct MessageMetrics (+ accessors)   kta::MessageMetrics   (view of a kta_result)
//   src/metric.rs:262-285 LogCompactionInMemoryMetrics         kta::LogCompactionInMemoryMetrics (view)
//   both `impl MetricHandler` (metric.rs:206, 288)             kta::HipMetricHandler: ONE handler feeds
//                                                              both reference handlers' state on the GPU
//
// The reference registers two handlers and calls each per message (kafka.rs:107-109); here one
// handler stages the message once and the device runs both accumulations.  Same accessor names,
// same integer semantics, same failure behaviour (the averages "panic" on divide by zero).
#pragma once

#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "kta_hip.h"

namespace kta {

// What the handlers read from a rdkafka BorrowedMessage (metric.rs:208-209, 218, 233).
struct Message {
    int32_t partition = 0;
    int64_t offset = 0;
    int64_t timestamp_ms = -1;     // raw rdkafka timestamp; -1 == not available
    const uint8_t *key = nullptr;  // nullptr == key None
    int64_t key_len = -1;
    int64_t payload_len = -1;      // -1 == payload None (tombstone); bytes are never read
};

struct RustPanic : std::runtime_error {
    std::string location;
    RustPanic(const std::string &msg, const std::string &loc) : std::runtime_error(msg), location(loc) {}
};

struct DateTimeUtc {  // chrono DateTime<Utc>, ordered by (sec, ns)
    int64_t sec = 0;
    uint32_t ns = 0;
    bool operator>(const DateTimeUtc &o) const { return sec > o.sec || (sec == o.sec && ns > o.ns); }
    bool operator<(const DateTimeUtc &o) const { return o > *this; }
};

class MetricHandler {  // kafka.rs:18-20
public:
    virtual ~MetricHandler() {}
    virtual void handle_message(const Message &m) = 0;
};

class MessageMetrics {  // accessors: metric.rs:104-195
public:
    MessageMetrics() {}
    MessageMetrics(const kta_result &r, std::vector<uint64_t> counters, DateTimeUtc now);
    uint64_t total(int32_t p) const { return metric(p, KTA_C_TOTAL); }
    uint64_t tombstones(int32_t p) const { return metric(p, KTA_C_TOMBSTONES); }
    uint64_t alive(int32_t p) const { return metric(p, KTA_C_ALIVE); }
    uint64_t key_null(int32_t p) const { return metric(p, KTA_C_KEY_NULL); }
    uint64_t key_non_null(int32_t p) const { return metric(p, KTA_C_KEY_NON_NULL); }
    uint64_t key_size_sum(int32_t p) const { return metric(p, KTA_C_KEY_SIZE_SUM); }
    uint64_t value_size_sum(int32_t p) const { return metric(p, KTA_C_VALUE_SIZE_SUM); }
    uint64_t key_size_avg(int32_t p) const;      // metric.rs:132-139 (throws RustPanic)
    uint64_t value_size_avg(int32_t p) const;    // metric.rs:141-148
    uint64_t message_size_avg(int32_t p) const;  // metric.rs:150-157
    float dirty_ratio(int32_t p) const;          // metric.rs:159-167
    const DateTimeUtc &latest_message() const { return latest_; }
    const DateTimeUtc &earliest_message() const { return earliest_; }
    uint64_t smallest_message() const;           // metric.rs:177-183
    uint64_t largest_message() const { return res_.largest_message; }
    uint64_t overall_count() const { return res_.overall_count; }
    uint64_t overall_size() const { return res_.overall_size; }

private:
    uint64_t metric(int32_t p, int c) const;     // metric.rs:198-203
    kta_result res_{};
    std::vector<uint64_t> c_;
    DateTimeUtc earliest_, latest_;
};

class LogCompactionInMemoryMetrics {  // metric.rs:262-285
public:
    LogCompactionInMemoryMetrics() {}
    explicit LogCompactionInMemoryMetrics(const kta_result &r) : alive_(r.alive_keys) {}
    size_t sum_all_alive() const { return (size_t)alive_; }

private:
    uint64_t alive_ = 0;
};

// The GPU-backed handler.  Construction == MessageMetrics::new() (+ LogCompactionInMemoryMetrics::new()
// when count_alive_keys): `now` stands in for Utc::now() (metric.rs:39).
class HipMetricHandler : public MetricHandler {
public:
    HipMetricHandler(int32_t n_partitions, bool count_alive_keys, int device = 0, uint64_t batch_capacity = 0,
                     uint64_t key_bytes_capacity = 0, uint32_t flags = 0);
    ~HipMetricHandler() override;
    HipMetricHandler(const HipMetricHandler &) = delete;
    HipMetricHandler &operator=(const HipMetricHandler &) = delete;

    void handle_message(const Message &m) override;  // kafka.rs:107-109
    // The trait has no end-of-stream hook; call this where main.rs:121 is (before the report).
    // tolerate_undelivered: records of batches that failed check.crcs / framing were written with
    // partition -1 (never counted); with true they are reported through undelivered_records()
    // instead of failing the run (the reference warns on Kafka errors and goes on, kafka.rs:95-97).
    void finish(bool tolerate_undelivered = false);
    // A rank of a partition-sharded run (one handler per GPU): join the job's communicator, and at the end
    // exchange() instead of finish() — afterwards metrics() / log_compaction() are the whole job's on every rank.
    void comm_create(int nranks, int rank, const uint8_t *unique_id);
    void exchange(bool tolerate_undelivered = false);
    uint64_t undelivered_records() const { return undelivered_; }
    const MessageMetrics &metrics() const { return metrics_; }
    const LogCompactionInMemoryMetrics *log_compaction() const { return alive_ ? &lc_ : nullptr; }
    kta_ctx *ctx() { return ctx_; }
    DateTimeUtc now() const { return now_; }

private:
    void check(int rc, const char *what);
    kta_ctx *ctx_ = nullptr;
    int32_t P_;
    bool alive_;
    DateTimeUtc now_;
    MessageMetrics metrics_;
    LogCompactionInMemoryMetrics lc_;
    uint64_t undelivered_ = 0;
};

// chrono 0.4.19 `Display for DateTime<Utc>` (main.rs:132-133)
std::string format_datetime_utc(int64_t sec, uint32_t ns);
// Rust `format!("{0:.4}", f32)` (main.rs:162)
std::string format_f32_4(float x);
// main.rs:123-178 — everything the reference prints after the scan, byte for byte
std::string render_report(const std::string &topic, uint64_t duration_secs, const MessageMetrics &m,
                          const LogCompactionInMemoryMetrics *lc, const std::vector<int32_t> &partitions,
                          const std::vector<int64_t> &start_offsets, const std::vector<int64_t> &end_offsets);

}  // namespace kta
