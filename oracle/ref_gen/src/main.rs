// Replays tests/golden/scenarios.json through the REFERENCE's own handlers (src/metric.rs, src/fnv32.rs,
// compiled from the reference checkout by path) and prints what its report would read.  See README.md.
extern crate bit_set;
extern crate chrono;
extern crate rdkafka;
extern crate serde_json;

use std::convert::TryInto;
use std::env;
use std::fs;
use std::panic;
use std::time::Instant;

use rdkafka::message::{BorrowedMessage, Timestamp};
use serde_json::{json, Value};

// what src/metric.rs imports from its own crate
pub mod kafka {
    use rdkafka::message::{BorrowedMessage, Message};
    pub trait MetricHandler {
        fn handle_message<'b>(&mut self, m: &BorrowedMessage<'b>) where BorrowedMessage<'b>: Message;
    }
}
include!(concat!(env!("OUT_DIR"), "/reference_paths.rs"));

use kafka::MetricHandler;
use metric::{LogCompactionInMemoryMetrics, MessageMetrics};

fn unhex(s: &str) -> Vec<u8> {
    (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).unwrap()).collect()
}

fn le_u32(b: &[u8], at: usize) -> u32 { u32::from_le_bytes(b[at..at + 4].try_into().unwrap()) }
fn le_i32(b: &[u8], at: usize) -> i32 { i32::from_le_bytes(b[at..at + 4].try_into().unwrap()) }
fn le_u64(b: &[u8], at: usize) -> u64 { u64::from_le_bytes(b[at..at + 8].try_into().unwrap()) }
fn le_i64(b: &[u8], at: usize) -> i64 { i64::from_le_bytes(b[at..at + 8].try_into().unwrap()) }
fn pad8(n: usize) -> usize { (n + 7) & !7 }

// `--time <topic.ktadump> [-c] [passes]`: the reference's own handle_message loop over a KTADUMP1 topic dump
// (kafka_topic_analyzer_amd/csrc/host/dump.hpp: what `kta-analyzer -b synthetic://c3 --librdkafka kta.write_dump=FILE`
// writes), timed: the true single-threaded CPU baseline ("kind": "reference" in bench.py's terms) on the very
// records the GPU path is fed.  Prints one JSON object: records, passes, seconds, records_per_s, and the
// report's totals of the last pass so that the run can be checked against the product's.
fn time_mode(path: &str, count_alive: bool, passes: usize) {
    let b = fs::read(path).expect("cannot read the dump");
    assert!(&b[0..8] == b"KTADUMP1" && le_u32(&b, 8) == 1, "not a KTADUMP1 file");
    let n_partitions = le_u32(&b, 12) as usize;
    let n_records = le_u64(&b, 16);
    let n_batches = le_u64(&b, 24) as usize;
    let batches_at = 32 + 16 * n_partitions;
    let mut best = f64::MAX;
    let mut last = json!(null);
    for _ in 0..passes {
        let mut mm = MessageMetrics::new();
        let mut lc = LogCompactionInMemoryMetrics::new();
        let value = vec![0u8; 1 << 20];                   // payload bytes are never read (src/metric.rs:233-245): one shared buffer
        let mut at = batches_at;
        let t0 = Instant::now();
        for _ in 0..n_batches {
            let n = le_u64(&b, at) as usize;
            let n_key_bytes = le_u64(&b, at + 8) as usize;
            let part = at + 16;
            let klen = part + pad8(4 * n);
            let vlen = klen + pad8(4 * n);
            let ts = vlen + pad8(4 * n);
            let koff = ts + pad8(8 * n);
            let keys = koff + pad8(4 * n);
            for i in 0..n {
                let kl = le_i32(&b, klen + 4 * i);
                let vl = le_i32(&b, vlen + 4 * i);
                let t = le_i64(&b, ts + 8 * i);
                let ko = keys + le_u32(&b, koff + 4 * i) as usize;
                let big;                                  // a value longer than the shared buffer (rare)
                let payload: Option<&[u8]> = if vl < 0 { None } else if (vl as usize) <= value.len() { Some(&value[..vl as usize]) }
                                             else { big = vec![0u8; vl as usize]; Some(&big[..]) };
                let m = BorrowedMessage {
                    partition: le_i32(&b, part + 4 * i),
                    timestamp: if t == -1 { Timestamp::NotAvailable } else { Timestamp::CreateTime(t) },
                    key: if kl < 0 { None } else { Some(&b[ko..ko + kl as usize]) },
                    payload: payload,
                };
                mm.handle_message(&m);                    // registration order of main.rs:108-115
                if count_alive { lc.handle_message(&m); }
            }
            at = keys + pad8(n_key_bytes);
        }
        let dt = t0.elapsed().as_secs_f64();
        if dt < best { best = dt; }
        last = json!({"overall_count": mm.overall_count(), "overall_size": mm.overall_size(),
                      "smallest": mm.smallest_message(), "largest": mm.largest_message(),
                      "alive_keys": if count_alive { json!(lc.sum_all_alive()) } else { json!(null) }});
    }
    println!("{}", serde_json::to_string(&json!({"kind": "reference", "cores": 1, "records": n_records, "passes": passes,
        "best_seconds": best, "records_per_s": n_records as f64 / best, "count_alive_keys": count_alive, "totals": last})).unwrap());
}

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() >= 3 && args[1] == "--time" {
        let count_alive = args.iter().any(|a| a == "-c");
        let passes = args.iter().skip(3).filter_map(|a| a.parse::<usize>().ok()).next().unwrap_or(3);
        time_mode(&args[2], count_alive, passes);
        return;
    }
    let path = env::args().nth(1).expect("usage: kta_ref_gen <scenarios.json> | --time <topic.ktadump> [-c] [passes]");
    let golden: Value = serde_json::from_str(&fs::read_to_string(path).unwrap()).unwrap();
    let n_partitions = golden["n_partitions"].as_i64().unwrap() as i32;
    let mut out = serde_json::Map::new();
    panic::set_hook(Box::new(|_| {}));   // the expected divide-by-zero panics are reported in the output, not on stderr
    for (name, sc) in golden["scenarios"].as_object().unwrap() {
        let mut mm = MessageMetrics::new();
        let mut lc = LogCompactionInMemoryMetrics::new();
        let created = *mm.earliest_message();   // Utc::now() at MessageMetrics::new (src/metric.rs:39)
        // records: [partition, ts_ms | null (not available), key hex | null, value length | null]
        for r in sc["records"].as_array().unwrap() {
            let key = r[2].as_str().map(unhex);
            let payload = r[3].as_u64().map(|n| vec![0u8; n as usize]);
            let m = BorrowedMessage {
                partition: r[0].as_i64().unwrap() as i32,
                timestamp: match r[1].as_i64() { Some(t) => Timestamp::CreateTime(t), None => Timestamp::NotAvailable },
                key: key.as_ref().map(|k| &k[..]),
                payload: payload.as_ref().map(|p| &p[..]),
            };
            mm.handle_message(&m);   // registration order of main.rs:108-115
            lc.handle_message(&m);
        }
        let mut parts = Vec::new();
        for p in 0..n_partitions {
            // the averages panic on "sum > 0 and alive == 0" (src/metric.rs:135, 144, 153)
            let avg = |f: &dyn Fn() -> u64| match panic::catch_unwind(panic::AssertUnwindSafe(|| f())) {
                Ok(v) => json!(v),
                Err(_) => json!("panic"),
            };
            parts.push(json!({
                "counters": [mm.total(p), mm.tombstones(p), mm.alive(p), mm.key_null(p), mm.key_non_null(p),
                             mm.key_size_sum(p), mm.value_size_sum(p)],
                "dirty_ratio_4": format!("{0:.4}", mm.dirty_ratio(p)),
                "key_size_avg": avg(&|| mm.key_size_avg(p)),
                "value_size_avg": avg(&|| mm.value_size_avg(p)),
                "message_size_avg": avg(&|| mm.message_size_avg(p)),
            }));
        }
        let earliest = *mm.earliest_message();
        out.insert(name.clone(), json!({
            "partitions": parts,
            "earliest": if earliest == created { json!("now") } else { json!([earliest.timestamp(), earliest.timestamp_subsec_nanos()]) },
            "latest": [mm.latest_message().timestamp(), mm.latest_message().timestamp_subsec_nanos()],
            "earliest_display": if earliest == created { json!("now") } else { json!(format!("{}", earliest)) },
            "latest_display": format!("{}", mm.latest_message()),
            "smallest": mm.smallest_message(),
            "largest": mm.largest_message(),
            "overall_count": mm.overall_count(),
            "overall_size": mm.overall_size(),
            "alive_keys": lc.sum_all_alive(),
        }));
    }
    println!("{}", serde_json::to_string_pretty(&Value::Object(out)).unwrap());
}
