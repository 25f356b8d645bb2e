// loop_driver.cpp — TEST HARNESS: runs kta::TopicAnalyzer (the reference's consume loop over librdkafka,
// host/rdkafka_source.cpp) with a recording MetricHandler, no GPU involved.  Prints one line per
// message the handlers saw and a summary; the Python test compares them with the synthetic topic.
//   loop_driver <bootstrap> <topic> [k=v,k=v]
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>

#include "rdkafka_source.hpp"

namespace {

struct Recorder : kta::MetricHandler {
    uint64_t n = 0;
    void handle_message(const kta::Message &m) override
    {
        uint32_t h = 0x811c9dc5u;   // any digest of the key bytes will do for the comparison
        for (int64_t i = 0; i < m.key_len; i++) h = (h ^ m.key[i]) * 0x01000193u;
        printf("M %d %lld %lld %lld %lld %u\n", m.partition, (long long)m.offset, (long long)m.timestamp_ms,
               (long long)m.key_len, (long long)m.payload_len, m.key ? h : 0u);
        n++;
    }
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    std::map<std::string, std::string> cfg;
    if (argc > 3) {
        std::string s = argv[3];
        size_t p0 = 0;
        while (p0 < s.size()) {
            size_t c = s.find(',', p0);
            std::string kv = s.substr(p0, c == std::string::npos ? std::string::npos : c - p0);
            size_t e = kv.find('=');
            cfg[kv.substr(0, e)] = kv.substr(e + 1);
            if (c == std::string::npos) break;
            p0 = c + 1;
        }
    }
    try {
        kta::TopicAnalyzer *ta = kta::TopicAnalyzer::new_from_bootstrap_servers(argv[1], cfg);
        std::map<int32_t, int64_t> start, end;
        ta->get_topic_offsets(argv[2], &start, &end);
        for (const auto &kv : end) printf("O %d %lld %lld\n", kv.first, (long long)start[kv.first], (long long)kv.second);
        Recorder first, second;     // two handlers, like the reference with -c: each sees every message
        ta->add_metric_handler(&first);
        ta->add_metric_handler(&second);
        const uint64_t seq = ta->read_topic_into_metrics(argv[2], end);
        delete ta;
        printf("S %llu %llu %llu\n", (unsigned long long)seq, (unsigned long long)first.n, (unsigned long long)second.n);
    } catch (const kta::RustPanic &p) {
        fprintf(stderr, "thread 'main' panicked at '%s', %s\n", p.what(), p.location.c_str());
        return 101;
    }
    return 0;
}
