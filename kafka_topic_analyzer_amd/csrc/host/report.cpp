// report.cpp — the report the reference prints after the scan (src/main.rs:123-178), byte for
// byte: fixed text block, chrono `DateTime<Utc>` Display, Rust `{:.4}` of an f32 and the
// prettytable-rs 0.8 default table format.  Also exported through the C ABI (kta_render_report)
// so that it can be tested without a GPU.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "metric.hpp"

namespace kta {

static void days_to_civil(int64_t z, int64_t *y, unsigned *m, unsigned *d)
{
    z += 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t yy = (int64_t)yoe + era * 400;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    *d = doy - (153 * mp + 2) / 5 + 1;
    *m = mp < 10 ? mp + 3 : mp - 9;
    *y = yy + (*m <= 2);
}

// chrono 0.4.19: `impl Display for DateTime<Tz>` writes "{naive_local} {offset}"; NaiveDate prints
// {:04}-{:02}-{:02} for years 0..=9999 (else {:+05}), NaiveTime prints {:02}:{:02}:{:02} followed by
// .{:03} / .{:06} / .{:09} when the nanosecond part is non-zero; Utc prints "UTC".
std::string format_datetime_utc(int64_t sec, uint32_t ns)
{
    int64_t days = sec / 86400, rem = sec % 86400;
    if (rem < 0) { rem += 86400; days -= 1; }
    int64_t y; unsigned mo, d;
    days_to_civil(days, &y, &mo, &d);
    char buf[96];
    int n;
    if (y >= 0 && y <= 9999) n = snprintf(buf, sizeof buf, "%04lld", (long long)y);
    else n = snprintf(buf, sizeof buf, "%+05lld", (long long)y);
    n += snprintf(buf + n, sizeof buf - n, "-%02u-%02u %02lld:%02lld:%02lld", mo, d, (long long)(rem / 3600),
                  (long long)((rem % 3600) / 60), (long long)(rem % 60));
    if (ns != 0) {
        if (ns % 1000000 == 0) n += snprintf(buf + n, sizeof buf - n, ".%03u", ns / 1000000);
        else if (ns % 1000 == 0) n += snprintf(buf + n, sizeof buf - n, ".%06u", ns / 1000);
        else n += snprintf(buf + n, sizeof buf - n, ".%09u", ns);
    }
    snprintf(buf + n, sizeof buf - n, " UTC");
    return buf;
}

// Rust formats the exact decimal expansion of the f32 rounded to 4 places; printf("%.4f") of the
// float promoted to double prints the same digits (the promotion is exact, glibc rounds the exact
// value half-to-even).
std::string format_f32_4(float x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.4f", (double)x);
    return buf;
}

// prettytable-rs 0.8.0, FORMAT_DEFAULT (what Table::new() + printstd() use): a `+---+` line above
// the first row and below every row, `|` column separators and borders, one space of padding on
// both sides, cells left-aligned.
static std::string pretty_table(const std::vector<std::vector<std::string>> &rows)
{
    size_t ncol = 0;
    for (auto &r : rows) ncol = std::max(ncol, r.size());
    std::vector<size_t> w(ncol, 0);
    for (auto &r : rows)
        for (size_t i = 0; i < r.size(); i++) w[i] = std::max(w[i], r[i].size());
    std::string sep = "+";
    for (size_t i = 0; i < ncol; i++) sep += std::string(w[i] + 2, '-') + "+";
    sep += "\n";
    std::string out = sep;
    for (auto &r : rows) {
        out += "|";
        for (size_t i = 0; i < ncol; i++) {
            const std::string &c = i < r.size() ? r[i] : std::string();
            out += " " + c + std::string(w[i] - c.size(), ' ') + " |";
        }
        out += "\n" + sep;
    }
    return out;
}

std::string render_report(const std::string &topic, uint64_t duration_secs, const MessageMetrics &m,
                          const LogCompactionInMemoryMetrics *lc, const std::vector<int32_t> &partitions,
                          const std::vector<int64_t> &start_offsets, const std::vector<int64_t> &end_offsets)
{
    auto u = [](uint64_t v) { return std::to_string(v); };
    const std::string eq(120, '='), dash(120, '-');
    std::string o;
    o += "\n";                                                                     // main.rs:125
    o += eq + "\n";                                                                // :126
    o += "Calculating statistics...\n";                                            // :127
    o += "Topic " + topic + "\n";                                                  // :128
    o += "Scanning took: " + u(duration_secs) + " seconds\n";                      // :129
    o += "Estimated Msg/s: " + u(m.overall_count() / std::max<uint64_t>(duration_secs, 1)) + "\n";  // :130
    o += dash + "\n";                                                              // :131
    o += "Earliest Message: " + format_datetime_utc(m.earliest_message().sec, m.earliest_message().ns) + "\n";
    o += "Latest Message: " + format_datetime_utc(m.latest_message().sec, m.latest_message().ns) + "\n";
    o += dash + "\n";                                                              // :134
    o += "Largest Message: " + u(m.largest_message()) + " bytes\n";                // :135
    o += "Smallest Message: " + u(m.smallest_message()) + " bytes\n";              // :136
    o += "Topic Size: " + u(m.overall_size()) + " bytes\n";                        // :137
    if (lc) {                                                                      // :139-146
        o += dash + "\n";
        o += "Alive keys: " + u(lc->sum_all_alive()) + "\n";
        o += dash + "\n";
    }
    o += eq + "\n";                                                                // :148
    std::vector<std::vector<std::string>> rows;
    rows.push_back({"P", "< OS", "> OS", "Total", "Alive", "Tmb", "DR", "K Null", "K !Null", "P-Bytes", "K-Bytes",
                    "V-Bytes", "A K-Sz", "A V-Sz", "A M-Sz"});                     // :150
    for (size_t i = 0; i < partitions.size(); i++) {                               // :153-172
        const int32_t p = partitions[i];
        const uint64_t key_size_avg = m.key_size_avg(p);  // first, as in main.rs:154 (may panic)
        rows.push_back({std::to_string(p), std::to_string(start_offsets[i]), std::to_string(end_offsets[i]),
                        u(m.total(p)), u(m.alive(p)), u(m.tombstones(p)), format_f32_4(m.dirty_ratio(p)),
                        u(m.key_null(p)), u(m.key_non_null(p)), u(m.key_size_sum(p) + m.value_size_sum(p)),
                        u(m.key_size_sum(p)), u(m.value_size_sum(p)), u(key_size_avg), u(m.value_size_avg(p)),
                        u(m.message_size_avg(p))});
    }
    o += "| K = Key, V = Value, P = Partition, Tmb = Tombstone(s), Sz = Size\n";    // :174
    o += "| DR = Dirty Ratio, A = Average, Lst = last, < OS = start offset, > OS = end offset\n";  // :175
    o += pretty_table(rows);                                                       // :176
    o += "\n";                                                                     // :177
    o += eq + "\n";                                                                // :178
    return o;
}

}  // namespace kta

extern "C" int kta_render_report(const char *topic, uint64_t duration_secs, const uint64_t *vec,
                                 uint32_t n_partitions, int count_alive_keys, int64_t now_sec, uint32_t now_ns,
                                 const int64_t *start_offsets, const int64_t *end_offsets, char *out,
                                 size_t out_cap, size_t *out_len)
{
    if (!topic || !vec || !out_len || n_partitions == 0) return KTA_ERR_INVALID;
    kta_result r;
    std::vector<uint64_t> counters((size_t)n_partitions * KTA_NCOUNTERS);
    int rc = kta_decode_vector(vec, n_partitions, count_alive_keys, &r, counters.data());
    if (rc != KTA_OK && rc != KTA_ERR_BAD_PARTITION) return rc;
    kta::MessageMetrics mm(r, std::move(counters), kta::DateTimeUtc{now_sec, now_ns});
    kta::LogCompactionInMemoryMetrics lc(r);
    std::vector<int32_t> parts(n_partitions);
    std::vector<int64_t> so(n_partitions), eo(n_partitions);
    for (uint32_t p = 0; p < n_partitions; p++) {
        parts[p] = (int32_t)p;
        so[p] = start_offsets ? start_offsets[p] : 0;
        eo[p] = end_offsets ? end_offsets[p] : (int64_t)mm.total((int32_t)p);
    }
    std::string text;
    try {
        text = kta::render_report(topic, duration_secs, mm, count_alive_keys ? &lc : nullptr, parts, so, eo);
    } catch (const kta::RustPanic &) {
        return KTA_ERR_DIV_BY_ZERO;  // the reference panics here (metric.rs:135,144,153)
    }
    *out_len = text.size();
    if (out && out_cap > 0) {
        const size_t n = std::min(out_cap - 1, text.size());
        memcpy(out, text.data(), n);
        out[n] = 0;
    }
    return KTA_OK;
}
