"""oracle_py.py — second, independent CPU restatement (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

A deliberately naive, pure-Python restatement of the reference hot path
(/root/reference/src/metric.rs, src/fnv32.rs) and of the report printer
(/root/reference/src/main.rs:121-179).  It exists to pin the C oracle
(oracle/kta_oracle.c): the Rust reference cannot be executed in the build image
and ships no tests, so two independently written restatements agreeing on
randomised inputs — plus the hand-derived known-answer vectors in tests/golden/ —
is the strongest pin available.  **Parity with the Rust binary itself stays
unpinned** (SURVEY.md §4, §8c).

Only tests/ may import this module.  Pure-Python loops: small inputs only.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Optional, Tuple

U32 = 0xFFFFFFFF
U64 = 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------- fnv32.rs
class FnvHasher:
    """fnv32.rs:74-101."""

    def __init__(self) -> None:  # fnv32.rs:79-81
        self.state = 0x811C9DC5

    def write(self, data: bytes) -> None:  # fnv32.rs:92-101
        h = self.state
        for byte in data:
            h = h ^ byte
            h = (h * 0x811C9DC5) & U32  # wrapping_mul; multiplier == offset basis
        self.state = h

    def finish(self) -> int:  # fnv32.rs:87-89
        return self.state


def fnv1a(data: bytes) -> int:  # metric.rs:256-260
    hasher = FnvHasher()
    hasher.write(data)
    return hasher.finish()


def fnv1a_standard(data: bytes) -> int:
    """The *standard* FNV-1a-32 (prime 0x01000193) — used only by tests that prove the
    reference variant is NOT the standard one."""
    h = 0x811C9DC5
    for byte in data:
        h = ((h ^ byte) * 0x01000193) & U32
    return h


# --------------------------------------------------------------------------- metric.rs
def _trunc_div(a: int, b: int) -> int:
    """Rust i64 `/`: truncates toward zero (Python's // floors)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


class DivideByZeroPanic(Exception):
    """Rust: thread 'main' panicked at 'attempt to divide by zero' (metric.rs:135,144,153)."""


class DateTimeRangePanic(Exception):
    """Rust: thread 'main' panicked at 'invalid or out-of-range datetime' — chrono 0.4.19 [3P]
    NaiveDateTime::from_timestamp (metric.rs:210, kafka.rs:104) outside NaiveDate's years
    [i32::MIN >> 13, i32::MAX >> 13] = [-262144, 262143]."""


def _days_from_civil(y: int, m: int, d: int) -> int:
    """Days from 1970-01-01 in the proleptic Gregorian calendar (year 0 = 1 BCE, as chrono counts)."""
    y -= m <= 2
    era = y // 400
    yoe = y - era * 400
    doy = (153 * (m + (-3 if m > 2 else 9)) + 2) // 5 + d - 1
    return era * 146097 + yoe * 365 + yoe // 4 - yoe // 100 + doy - 719468


CHRONO_MIN_SEC = _days_from_civil(-262144, 1, 1) * 86400            # -8 334 632 851 200
CHRONO_MAX_SEC = _days_from_civil(262143, 12, 31) * 86400 + 86399   # 8 210 298 412 799


def _f32(x: float) -> float:
    return struct.unpack("<f", struct.pack("<f", x))[0]


class MessageMetrics:
    """metric.rs:11-253.  Timestamps are (sec, ns) tuples ordered lexicographically."""

    def __init__(self, now: Tuple[int, int]) -> None:  # metric.rs:30-46
        self.total_messages: Dict[int, int] = {}
        self.tombstones_: Dict[int, int] = {}
        self.alive_: Dict[int, int] = {}
        self.key_null_: Dict[int, int] = {}
        self.key_non_null_: Dict[int, int] = {}
        self.key_size_sum_: Dict[int, int] = {}
        self.value_size_sum_: Dict[int, int] = {}
        self.earliest_message = now
        self.latest_message = (0, 0)
        self.largest_message_ = 0
        self.smallest_message_ = U64
        self.overall_size_ = 0
        self.overall_count_ = 0

    @staticmethod
    def _inc(bucket: Dict[int, int], p: int, amount: int = 1) -> None:  # metric.rs:74-100
        bucket[p] = (bucket.get(p, 0) + amount) & U64

    def handle_message(self, partition: int, timestamp_ms: Optional[int],
                       key: Optional[bytes], payload_len: Optional[int]) -> None:
        """metric.rs:207-252.  timestamp_ms None == Timestamp::to_millis() None."""
        if timestamp_ms is None or timestamp_ms == -1:  # rdkafka 0.25 to_millis
            timestamp_ms = 0  # unwrap_or(0)  :209
        timestamp_dt = (_trunc_div(timestamp_ms, 1000), 0)  # :210-211
        if not CHRONO_MIN_SEC <= timestamp_dt[0] <= CHRONO_MAX_SEC:
            raise DateTimeRangePanic(timestamp_ms)  # .expect("invalid or out-of-range datetime") [3P chrono]
        message_size = 0
        empty_value = False
        self.overall_count_ += 1  # :215
        self._inc(self.total_messages, partition)  # :216
        if key is not None:  # :219-226
            self._inc(self.key_non_null_, partition)
            message_size += len(key)
            self._inc(self.key_size_sum_, partition, len(key))
            self.overall_size_ += len(key)
        else:  # :227-230
            self._inc(self.key_null_, partition)
        if payload_len is not None:  # :234-240
            message_size += payload_len
            self._inc(self.value_size_sum_, partition, payload_len)
            self.overall_size_ += payload_len
            self._inc(self.alive_, partition)
        else:  # :241-244
            empty_value = True
            self._inc(self.tombstones_, partition)
        # :247 -> :65-72
        if self.earliest_message > timestamp_dt:
            self.earliest_message = timestamp_dt
        if self.latest_message < timestamp_dt:
            self.latest_message = timestamp_dt
        if not empty_value:  # :249-251 -> :56-63
            if self.largest_message_ < message_size:
                self.largest_message_ = message_size
            if self.smallest_message_ > message_size:
                self.smallest_message_ = message_size

    # accessors metric.rs:104-130
    def total(self, p: int) -> int: return self.total_messages.get(p, 0)
    def tombstones(self, p: int) -> int: return self.tombstones_.get(p, 0)
    def alive(self, p: int) -> int: return self.alive_.get(p, 0)
    def key_null(self, p: int) -> int: return self.key_null_.get(p, 0)
    def key_non_null(self, p: int) -> int: return self.key_non_null_.get(p, 0)
    def key_size_sum(self, p: int) -> int: return self.key_size_sum_.get(p, 0)
    def value_size_sum(self, p: int) -> int: return self.value_size_sum_.get(p, 0)

    def _avg(self, s: int, p: int) -> int:  # metric.rs:132-157
        if s > 0:
            if self.alive(p) == 0:
                raise DivideByZeroPanic()
            return s // self.alive(p)
        return 0

    def key_size_avg(self, p: int) -> int: return self._avg(self.key_size_sum(p), p)
    def value_size_avg(self, p: int) -> int: return self._avg(self.value_size_sum(p), p)

    def message_size_avg(self, p: int) -> int:
        return self._avg(self.key_size_sum(p) + self.value_size_sum(p), p)

    def dirty_ratio(self, p: int) -> float:  # metric.rs:159-167 (f32)
        t, tm = self.tombstones(p), self.total(p)
        if tm > 0 and t > 0:
            return _f32(_f32(float(t)) / _f32(_f32(float(tm)) / _f32(100.0)))
        return 0.0

    def smallest_message(self) -> int:  # metric.rs:177-183
        return 0 if self.smallest_message_ == U64 else self.smallest_message_

    def largest_message(self) -> int: return self.largest_message_
    def overall_count(self) -> int: return self.overall_count_
    def overall_size(self) -> int: return self.overall_size_

    def counters(self, n_partitions: int) -> List[int]:
        out: List[int] = []
        for p in range(n_partitions):
            out += [self.total(p), self.tombstones(p), self.alive(p), self.key_null(p),
                    self.key_non_null(p), self.key_size_sum(p), self.value_size_sum(p)]
        return out


class LogCompactionInMemoryMetrics:
    """metric.rs:262-305; BitSet modelled as a Python set of usize."""

    def __init__(self) -> None:
        self.store = set()

    def mark_key_alive(self, key: bytes) -> None: self.store.add(fnv1a(key))  # :273-276
    def mark_key_dead(self, key: bytes) -> None: self.store.discard(fnv1a(key))  # :278-280
    def sum_all_alive(self) -> int: return len(self.store)  # :282-284

    def handle_message(self, key: Optional[bytes], payload_len: Optional[int]) -> None:  # :289-304
        if key is not None:
            if payload_len is not None:
                self.mark_key_alive(key)
            else:
                self.mark_key_dead(key)


Record = Tuple[int, Optional[int], Optional[bytes], Optional[int]]


def run(records: Iterable[Record], now: Tuple[int, int], count_alive_keys: bool):
    """kafka.rs:107-109 with the registration order of main.rs:108-115."""
    mm = MessageMetrics(now)
    lc = LogCompactionInMemoryMetrics() if count_alive_keys else None
    for part, ts, key, vlen in records:
        mm.handle_message(part, ts, key, vlen)
        if lc is not None:
            lc.handle_message(key, vlen)
    return mm, lc


# --------------------------------------------------------------------------- main.rs report
def _days_to_civil(z: int) -> Tuple[int, int, int]:
    z += 719468
    era = (z if z >= 0 else z - 146096) // 146097
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = mp + 3 if mp < 10 else mp - 9
    return (y + 1 if m <= 2 else y, m, d)


def format_datetime_utc(sec: int, ns: int) -> str:
    """chrono 0.4.19 `impl Display for DateTime<Utc>`: "{naive_local} {offset}" where the
    NaiveDateTime prints `%Y-%m-%d %H:%M:%S` followed by `.fff`, `.ffffff` or `.fffffffff`
    when the nanosecond part is non-zero (shortest of 3/6/9 digits that is exact), and the
    Utc offset prints "UTC"."""
    days, rem = divmod(sec, 86400)
    y, mo, d = _days_to_civil(days)
    hh, rem = divmod(rem, 3600)
    mi, ss = divmod(rem, 60)
    if 0 <= y <= 9999:
        ys = "%04d" % y
    else:
        ys = "%+05d" % y
    s = "%s-%02d-%02d %02d:%02d:%02d" % (ys, mo, d, hh, mi, ss)
    if ns:
        if ns % 1_000_000 == 0:
            s += ".%03d" % (ns // 1_000_000)
        elif ns % 1_000 == 0:
            s += ".%06d" % (ns // 1_000)
        else:
            s += ".%09d" % ns
    return s + " UTC"


def format_f32_4(x: float) -> str:
    """Rust `format!("{0:.4}", f32)`: exact decimal expansion of the f32, rounded to 4
    places, ties-to-even on the exact value (matches Python's Decimal-exact formatting)."""
    from decimal import Decimal, ROUND_HALF_EVEN
    return str(Decimal(_f32(x)).quantize(Decimal("0.0001"), rounding=ROUND_HALF_EVEN))


def prettytable(rows: List[List[str]]) -> str:
    """prettytable-rs 0.8.0 `Table::printstd()` with the default format
    (FORMAT_DEFAULT): `+---+` separator above the first row and below every row, cells
    left-aligned with one space of padding on each side, `|` borders."""
    ncol = max(len(r) for r in rows)
    widths = [0] * ncol
    for r in rows:
        for i, c in enumerate(r):
            widths[i] = max(widths[i], len(c))
    sep = "+" + "+".join("-" * (w + 2) for w in widths) + "+\n"
    out = sep
    for r in rows:
        out += "|" + "|".join(" " + c + " " * (widths[i] - len(c)) + " "
                              for i, c in enumerate(r)) + "|\n"
        out += sep
    return out


HEADER = ["P", "< OS", "> OS", "Total", "Alive", "Tmb", "DR", "K Null", "K !Null", "P-Bytes",
          "K-Bytes", "V-Bytes", "A K-Sz", "A V-Sz", "A M-Sz"]


def report(topic: str, duration_secs: int, mm: MessageMetrics,
           lc: Optional[LogCompactionInMemoryMetrics], partitions: List[int],
           start_offsets: Dict[int, int], end_offsets: Dict[int, int],
           os_names: Tuple[str, str] = ("< OS", "> OS")) -> str:
    """main.rs:123-179 (stdout only).  `os_names`: the two offset column names — v0.5.0 prints
    "< OS" / "> OS" (main.rs:150,175); the build behind /root/reference/demo_output.png printed
    "|< OS" / ">| OS" (tests/test_reference_demo.py compares against that screenshot)."""
    o = "\n"
    o += "=" * 120 + "\n"
    o += "Calculating statistics...\n"
    o += "Topic %s\n" % topic
    o += "Scanning took: %d seconds\n" % duration_secs
    o += "Estimated Msg/s: %d\n" % (mm.overall_count() // max(duration_secs, 1))
    o += "-" * 120 + "\n"
    o += "Earliest Message: %s\n" % format_datetime_utc(*mm.earliest_message)
    o += "Latest Message: %s\n" % format_datetime_utc(*mm.latest_message)
    o += "-" * 120 + "\n"
    o += "Largest Message: %d bytes\n" % mm.largest_message()
    o += "Smallest Message: %d bytes\n" % mm.smallest_message()
    o += "Topic Size: %d bytes\n" % mm.overall_size()
    if lc is not None:
        o += "-" * 120 + "\n"
        o += "Alive keys: %d\n" % lc.sum_all_alive()
        o += "-" * 120 + "\n"
    o += "=" * 120 + "\n"
    rows = [[HEADER[0], os_names[0], os_names[1]] + HEADER[3:]]
    for p in sorted(partitions):
        key_size_avg = mm.key_size_avg(p)  # may raise DivideByZeroPanic (main.rs:154)
        rows.append([
            str(p), str(start_offsets[p]), str(end_offsets[p]), str(mm.total(p)),
            str(mm.alive(p)), str(mm.tombstones(p)), format_f32_4(mm.dirty_ratio(p)),
            str(mm.key_null(p)), str(mm.key_non_null(p)),
            str(mm.key_size_sum(p) + mm.value_size_sum(p)), str(mm.key_size_sum(p)),
            str(mm.value_size_sum(p)), str(key_size_avg), str(mm.value_size_avg(p)),
            str(mm.message_size_avg(p)),
        ])
    o += "| K = Key, V = Value, P = Partition, Tmb = Tombstone(s), Sz = Size\n"
    o += "| DR = Dirty Ratio, A = Average, Lst = last, %s = start offset, %s = end offset\n" % os_names
    o += prettytable(rows)
    o += "\n"
    o += "=" * 120 + "\n"
    return o
