//! Stand-in for the `rdkafka` crate (0.25.0): the message surface `src/metric.rs` reads, nothing else.
pub mod message {
    /// rdkafka 0.25.0 `Timestamp`: `to_millis` is `None` for `NotAvailable` and for a value of -1.
    #[derive(Clone, Copy, Debug)]
    pub enum Timestamp {
        NotAvailable,
        CreateTime(i64),
        LogAppendTime(i64),
    }

    impl Timestamp {
        pub fn to_millis(self) -> Option<i64> {
            match self {
                Timestamp::NotAvailable | Timestamp::CreateTime(-1) | Timestamp::LogAppendTime(-1) => None,
                Timestamp::CreateTime(t) | Timestamp::LogAppendTime(t) => Some(t),
            }
        }
    }

    /// The accessors the handlers call (src/metric.rs:208-209, 218, 233, 291-293).
    pub trait Message {
        fn key(&self) -> Option<&[u8]>;
        fn payload(&self) -> Option<&[u8]>;
        fn partition(&self) -> i32;
        fn timestamp(&self) -> Timestamp;
    }

    /// The real type borrows from librdkafka; this one borrows from the generator's buffers.
    pub struct BorrowedMessage<'a> {
        pub partition: i32,
        pub timestamp: Timestamp,
        pub key: Option<&'a [u8]>,
        pub payload: Option<&'a [u8]>,
    }

    impl<'a> Message for BorrowedMessage<'a> {
        fn key(&self) -> Option<&[u8]> { self.key }
        fn payload(&self) -> Option<&[u8]> { self.payload }
        fn partition(&self) -> i32 { self.partition }
        fn timestamp(&self) -> Timestamp { self.timestamp }
    }
}
