// rdkafka_source.hpp — the reference's consume loop (src/kafka.rs:13-137) over librdkafka's C API.
//
//   reference                                             here
//   struct TopicAnalyzer<'a> (kafka.rs:13-16)             kta::TopicAnalyzer
//   new_from_bootstrap_servers (kafka.rs:23-54)           TopicAnalyzer::new_from_bootstrap_servers
//   add_metric_handler (kafka.rs:56-58)                   TopicAnalyzer::add_metric_handler
//   get_topic_offsets (kafka.rs:60-72)                    TopicAnalyzer::get_topic_offsets
//   read_topic_into_metrics (kafka.rs:74-137)             TopicAnalyzer::read_topic_into_metrics
//
// The reference reaches librdkafka 1.6.0 through the rdkafka crate (Cargo.lock:595-613).  This build
// image has neither the library nor its header, so the C API is bound at RUN time: the shared object
// ($KTA_RDKAFKA_LIB, else librdkafka.so.1, else librdkafka.so) is dlopen()ed and the ~20 entry points
// used below are resolved by name; the two public structs read here (rd_kafka_message_t and the
// metadata structs) are declared in rdkafka_source.cpp from librdkafka's documented, ABI-stable
// layout.  Nothing is needed at compile time and the analyzer still runs without the library for
// every other record source.  tests/mock_rdkafka.c implements the same entry points over the
// synthetic topic, which is how the loop is tested here.
#pragma once

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "metric.hpp"

namespace kta {

class TopicAnalyzer {
public:
    // kafka.rs:23-54.  Throws RustPanic("Consumer creation failed: ...") like the reference's expect().
    // Keys starting with "kta." are this build's own knobs and are not forwarded.
    static TopicAnalyzer *new_from_bootstrap_servers(const std::string &bootstrap_server,
                                                     const std::map<std::string, std::string> &librdkafka_settings);
    ~TopicAnalyzer();
    TopicAnalyzer(const TopicAnalyzer &) = delete;
    TopicAnalyzer &operator=(const TopicAnalyzer &) = delete;

    void add_metric_handler(MetricHandler *handler) { metric_handlers_.push_back(handler); }   // kafka.rs:56-58

    // kafka.rs:60-72: metadata of `topic`, then low / high watermark of every partition.
    void get_topic_offsets(const std::string &topic, std::map<int32_t, int64_t> *start_offsets,
                           std::map<int32_t, int64_t> *end_offsets);

    // kafka.rs:74-137: subscribe, poll(100 ms) until every partition delivered offset + 1 >= end.
    // Returns the number of messages handed to the handlers (the reference's `seq`).
    uint64_t read_topic_into_metrics(const std::string &topic, const std::map<int32_t, int64_t> &end_offsets);

private:
    TopicAnalyzer() {}
    struct Api;
    Api *api_ = nullptr;
    void *rk_ = nullptr;   // rd_kafka_t*
    std::vector<MetricHandler *> metric_handlers_;
};

}  // namespace kta
