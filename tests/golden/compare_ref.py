"""Compare the output of oracle/ref_gen (the REFERENCE's own src/metric.rs + src/fnv32.rs replaying the golden
inputs; needs a box with cargo) with tests/golden/scenarios.json, which the oracle is tested against.
    python tests/golden/compare_ref.py ref_scenarios.json
Exit code 0 and "pinned" when every value agrees: commit the file as tests/golden/ref_scenarios.json then."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
ref = json.load(open(sys.argv[1]))
gold = json.load(open(os.path.join(here, "scenarios.json")))
now = gold["now"]
bad = 0
for name, sc in gold["scenarios"].items():
    want, got = sc["expect"], ref[name]
    for key in ("smallest", "largest", "overall_count", "overall_size", "alive_keys", "latest", "latest_display"):
        if want[key] != got[key]:
            bad += 1
            print(f"{name}.{key}: golden {want[key]!r} != reference {got[key]!r}")
    w_earliest = "now" if want["earliest"] == now else want["earliest"]     # the golden file substitutes a fixed "now"
    if w_earliest != got["earliest"]:
        bad += 1
        print(f"{name}.earliest: golden {w_earliest!r} != reference {got['earliest']!r}")
    for p, (wp, gp) in enumerate(zip(want["partitions"], got["partitions"])):
        for key in ("counters", "dirty_ratio_4", "key_size_avg", "value_size_avg", "message_size_avg"):
            if wp[key] != gp[key]:
                bad += 1
                print(f"{name}.partitions[{p}].{key}: golden {wp[key]!r} != reference {gp[key]!r}")
print("pinned: the reference agrees with every golden value" if not bad else f"{bad} differences")
sys.exit(1 if bad else 0)
