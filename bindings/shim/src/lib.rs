//! `databend-common-hip`: Databend's column-batch hot path on one MI355X behind the reference's own traits.
//! See ../README.md for what each module implements and why this crate is source only.
#![allow(clippy::missing_safety_doc)]

/// The raw FFI (generated from include/dbhip.h by tools/gen_rust_bindings.py; kept one directory up).
#[path = "../../dbhip_sys.rs"]
pub mod sys;

pub mod aggregate;
pub mod device;
pub mod join;
pub mod scalar;

use databend_common_exception::ErrorCode;
use databend_common_exception::Result;

/// dbhip status -> the reference's ErrorCode (the same mapping as `check()` of the C++ mirror, dbhip_host.hpp:45-52).
pub fn check(rc: i32) -> Result<()> {
    if rc == sys::DBHIP_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(sys::dbhip_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        sys::DBHIP_ERR_OVERFLOW => ErrorCode::Overflow(msg),
        sys::DBHIP_ERR_UNSUPPORTED => ErrorCode::Unimplemented(msg),
        sys::DBHIP_ERR_INVALID => ErrorCode::BadArguments(msg),
        sys::DBHIP_ERR_CANCELLED => ErrorCode::AbortedQuery(msg),
        _ => ErrorCode::Internal(msg),
    })
}

/// Statuses after which NOTHING was changed on the device and the CPU operator takes the block (or the rest of the stream).
pub fn falls_back(rc: i32) -> bool {
    rc == sys::DBHIP_ERR_UNSUPPORTED || rc == sys::DBHIP_ERR_CAPACITY || rc == sys::DBHIP_ERR_ROW_ERRORS
}
