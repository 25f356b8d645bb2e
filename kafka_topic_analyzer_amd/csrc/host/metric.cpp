// metric.cpp — host-side accessors and the GPU-backed MetricHandler (see metric.hpp).
#include "metric.hpp"

#include <time.h>

namespace kta {

MessageMetrics::MessageMetrics(const kta_result &r, std::vector<uint64_t> counters, DateTimeUtc now)
    : res_(r), c_(std::move(counters))
{
    // metric.rs:39-40: earliest = Utc::now(), latest = epoch 0; then what
    // cmp_and_set_message_timestamp (metric.rs:65-72) would have done with the device extrema.
    earliest_ = now;
    latest_ = DateTimeUtc{0, 0};
    if (r.any_records) {
        DateTimeUtc lo{r.min_ts_sec, 0}, hi{r.max_ts_sec, 0};
        if (earliest_ > lo) earliest_ = lo;
        if (latest_ < hi) latest_ = hi;
    }
}

uint64_t MessageMetrics::metric(int32_t p, int c) const
{
    if (p < 0 || (uint32_t)p >= res_.n_partitions) return 0;  // None => 0 (metric.rs:201)
    return c_[(size_t)p * KTA_NCOUNTERS + c];
}

static uint64_t avg_or_panic(uint64_t sum, uint64_t alive, const char *loc)
{
    if (sum > 0) {
        if (alive == 0) throw RustPanic("attempt to divide by zero", loc);
        return sum / alive;
    }
    return 0;
}

uint64_t MessageMetrics::key_size_avg(int32_t p) const
{
    return avg_or_panic(key_size_sum(p), alive(p), "src/metric.rs:135:13");
}

uint64_t MessageMetrics::value_size_avg(int32_t p) const
{
    return avg_or_panic(value_size_sum(p), alive(p), "src/metric.rs:144:13");
}

uint64_t MessageMetrics::message_size_avg(int32_t p) const
{
    return avg_or_panic(key_size_sum(p) + value_size_sum(p), alive(p), "src/metric.rs:153:13");
}

float MessageMetrics::dirty_ratio(int32_t p) const
{
    const uint64_t total_messages = total(p), tomb = tombstones(p);
    if (total_messages > 0 && tomb > 0) {
        // two f32 roundings, exactly as `tombstones as f32 / (total_messages as f32 / 100.0f32)`;
        // compiled with -ffp-contract=off, volatile keeps each step in f32
        volatile float t = (float)tomb;
        volatile float tm = (float)total_messages;
        volatile float d = tm / 100.0f;
        volatile float r = t / d;
        return r;
    }
    return 0.0f;
}

uint64_t MessageMetrics::smallest_message() const
{
    return res_.smallest_message == UINT64_MAX ? 0 : res_.smallest_message;
}

HipMetricHandler::HipMetricHandler(int32_t n_partitions, bool count_alive_keys, int device, uint64_t batch_capacity,
                                   uint64_t key_bytes_capacity, uint32_t flags)
    : P_(n_partitions), alive_(count_alive_keys)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);  // Utc::now() (metric.rs:39)
    now_ = DateTimeUtc{(int64_t)ts.tv_sec, (uint32_t)ts.tv_nsec};
    kta_config cfg{};
    cfg.device_id = device;
    cfg.n_partitions = n_partitions;
    cfg.count_alive_keys = count_alive_keys ? 1 : 0;
    cfg.batch_capacity = batch_capacity;
    cfg.key_bytes_capacity = key_bytes_capacity;
    cfg.flags = flags;
    int rc = kta_create(&cfg, &ctx_);
    if (rc != KTA_OK) throw std::runtime_error(std::string("kta_create failed: ") + kta_last_error(nullptr));
}

HipMetricHandler::~HipMetricHandler() { kta_destroy(ctx_); }

void HipMetricHandler::check(int rc, const char *what)
{
    if (rc != KTA_OK) throw std::runtime_error(std::string(what) + ": " + kta_last_error(ctx_));
}

// NaiveDateTime::from_timestamp(timestamp / 1000, 0) of a record outside chrono 0.4.19's range: the reference
// dies on that record (kafka.rs:104, before any handler; metric.rs:210 holds the same call).  The device counts the
// record like any other; the extrema tell at the end of the stream, which is where this mirror can say it.  (The
// reference's own location string is the expect() inside the chrono crate, under the builder's cargo registry.)
static const char *const kDateTimePanic = "invalid or out-of-range datetime";
static const char *const kDateTimePanicAt = "chrono-0.4.19/src/naive/datetime.rs (NaiveDateTime::from_timestamp, src/kafka.rs:104)";

void HipMetricHandler::handle_message(const Message &m)
{
    // metric.rs:209-210: the handler's own NaiveDateTime::from_timestamp(timestamp / 1000, 0) — the message is in
    // hand here, so a timestamp outside chrono's range ends the run on this very record, as in the reference
    const int64_t secs = (m.timestamp_ms == -1 ? 0 : m.timestamp_ms) / 1000;
    if (secs < KTA_CHRONO_MIN_SEC || secs > KTA_CHRONO_MAX_SEC)
        throw RustPanic(kDateTimePanic, "chrono-0.4.19/src/naive/datetime.rs (NaiveDateTime::from_timestamp, src/metric.rs:210)");
    check(kta_handle_message(ctx_, m.partition, m.timestamp_ms, m.key, m.key ? m.key_len : -1, m.payload_len),
          "kta_handle_message");
}

void HipMetricHandler::finish(bool tolerate_undelivered)
{
    kta_result r{};
    std::vector<uint64_t> counters((size_t)P_ * KTA_NCOUNTERS);
    const int rc = kta_finish(ctx_, &r, counters.data());
    if (rc == KTA_ERR_TIMESTAMP_RANGE) throw RustPanic(kDateTimePanic, kDateTimePanicAt);
    if (!(rc == KTA_ERR_BAD_PARTITION && tolerate_undelivered)) check(rc, "kta_finish");
    undelivered_ = r.bad_partition_records;
    metrics_ = MessageMetrics(r, std::move(counters), now_);
    lc_ = LogCompactionInMemoryMetrics(r);
}

void HipMetricHandler::comm_create(int nranks, int rank, const uint8_t *unique_id)
{
    check(kta_comm_create(ctx_, nranks, rank, unique_id), "kta_comm_create");
}

void HipMetricHandler::exchange(bool tolerate_undelivered)
{
    check(kta_exchange(ctx_), "kta_exchange");
    kta_result r{};
    std::vector<uint64_t> counters((size_t)P_ * KTA_NCOUNTERS);
    const int rc = kta_exchange_result(ctx_, &r, counters.data());
    if (rc == KTA_ERR_TIMESTAMP_RANGE) throw RustPanic(kDateTimePanic, kDateTimePanicAt);
    if (!(rc == KTA_ERR_BAD_PARTITION && tolerate_undelivered)) check(rc, "kta_exchange_result");
    undelivered_ = r.bad_partition_records;
    metrics_ = MessageMetrics(r, std::move(counters), now_);
    lc_ = LogCompactionInMemoryMetrics(r);
}

}  // namespace kta
