// Writes $OUT_DIR/reference_paths.rs: the two reference modules, included by path (never copied).
use std::env;
use std::fs;
use std::path::Path;

fn main() {
    let root = env::var("KTA_REFERENCE").unwrap_or_else(|_| "/root/reference".to_string());
    let out = env::var("OUT_DIR").unwrap();
    let text = format!(
        "#[path = \"{0}/src/fnv32.rs\"]\npub mod fnv32;\n#[path = \"{0}/src/metric.rs\"]\npub mod metric;\n",
        root
    );
    fs::write(Path::new(&out).join("reference_paths.rs"), text).unwrap();
    println!("cargo:rerun-if-env-changed=KTA_REFERENCE");
}
