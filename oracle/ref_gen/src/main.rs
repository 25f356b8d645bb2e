// Replays tests/golden/scenarios.json through the REFERENCE's own handlers (src/metric.rs, src/fnv32.rs,
// compiled from the reference checkout by path) and prints what its report would read.  See README.md.
extern crate bit_set;
extern crate chrono;
extern crate rdkafka;
extern crate serde_json;

use std::env;
use std::fs;
use std::panic;

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

fn main() {
    let path = env::args().nth(1).expect("usage: kta_ref_gen <scenarios.json>");
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
