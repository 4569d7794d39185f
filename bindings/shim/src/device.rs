//! Column hand-over. A `DeviceColumn` is a reference `Column` whose bytes live in HBM: uploaded once per block
//! with `dbhip_memcpy_h2d` (57 GB/s measured, profiles/r05_h2d_rate.json) or — the intended deployment — produced on the device by
//! the scan (`dbhip_pq_chunks_decode_device`) and carried through the pipeline inside `Column::Opaque`.
use std::ffi::c_void;
use std::ptr::null;

use databend_common_column::bitmap::Bitmap;
use databend_common_column::buffer::Buffer;
use databend_common_exception::Result;
use databend_common_expression::types::AnyType;
use databend_common_expression::types::DataType;
use databend_common_expression::types::DecimalSize;
use databend_common_expression::types::NumberDataType;
use databend_common_expression::BlockEntry;
use databend_common_expression::Column;
use databend_common_expression::Scalar;
use databend_common_expression::Value;

use crate::check;
use crate::sys::*;

/// RAII over `dbhip_alloc` / `dbhip_free` (the library's block cache: no hipMalloc per block).
pub struct DeviceBuffer {
    ptr: *mut c_void,
    bytes: usize,
}
unsafe impl Send for DeviceBuffer {}
unsafe impl Sync for DeviceBuffer {}

impl DeviceBuffer {
    pub fn alloc(bytes: usize) -> Result<Self> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { dbhip_alloc(bytes.max(16) + 16, &mut p) })?;
        Ok(Self { ptr: p, bytes })
    }
    pub fn upload<T>(src: &[T], stream: *mut c_void) -> Result<Self> {
        let b = Self::alloc(std::mem::size_of_val(src))?;
        if !src.is_empty() {
            check(unsafe { dbhip_memcpy_h2d(b.ptr, src.as_ptr() as *const c_void, std::mem::size_of_val(src), stream) })?;
        }
        Ok(b)
    }
    pub fn download<T: Default + Clone>(&self, n: usize, stream: *mut c_void) -> Result<Vec<T>> {
        let mut v = vec![T::default(); n];
        if n > 0 {
            check(unsafe { dbhip_memcpy_d2h(v.as_mut_ptr() as *mut c_void, self.ptr, n * std::mem::size_of::<T>(), stream) })?;
        }
        Ok(v)
    }
    pub fn ptr(&self) -> *mut c_void { self.ptr }
    pub fn bytes(&self) -> usize { self.bytes }
}
impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        unsafe { dbhip_free(self.ptr) };
    }
}

/// One column of a block as the C-ABI sees it (`dbhip_col`, include/dbhip.h:97-108) plus the buffers that keep it alive.
pub struct DeviceColumn {
    pub data_type: DataType,
    pub len: usize,
    pub dbhip_type: i32,
    pub is_scalar: bool,
    pub data: DeviceBuffer,
    pub validity: Option<(DeviceBuffer, usize)>, // (bitmap bytes, bit offset)
    pub size: DecimalSize,
}

impl DeviceColumn {
    pub fn as_col(&self) -> dbhip_col {
        dbhip_col {
            r#type: self.dbhip_type,
            is_scalar: self.is_scalar as i32,
            data: self.data.ptr() as *const c_void,
            validity: self.validity.as_ref().map_or(null(), |(b, _)| b.ptr() as *const u8),
            validity_offset: self.validity.as_ref().map_or(0, |(_, off)| *off as i64),
            buffers: null(),
            n_buffers: 0,
            precision: self.size.precision,
            scale: self.size.scale,
            _pad: [0; 2],
        }
    }

    /// `BlockEntry` (block.rs:55-59) -> device. `Const` entries cross as ONE value with `is_scalar = 1` (values.rs:122).
    pub fn from_entry(entry: &BlockEntry, stream: *mut c_void) -> Result<Option<Self>> {
        match entry {
            BlockEntry::Column(c) => Self::from_column(c, stream),
            BlockEntry::Const(s, ty, n) => Self::from_scalar(s, ty, *n, stream),
        }
    }

    pub fn from_value(v: &Value<AnyType>, ty: &DataType, n: usize, stream: *mut c_void) -> Result<Option<Self>> {
        match v {
            Value::Column(c) => Self::from_column(c, stream),
            Value::Scalar(s) => Self::from_scalar(s, ty, n, stream),
        }
    }

    /// None = a type outside the device path (the caller keeps the CPU operator for the block).
    pub fn from_column(c: &Column, stream: *mut c_void) -> Result<Option<Self>> {
        let (inner, validity): (&Column, Option<&Bitmap>) = match c {
            Column::Nullable(n) => (&n.column, Some(&n.validity)),
            other => (other, None),
        };
        let up = |t: i32, bytes: &[u8], size: DecimalSize| -> Result<Option<Self>> {
            let validity = match validity {
                Some(b) => {
                    let (slice, off, _len) = b.as_slice(); // bitmap/immutable.rs:78-85
                    Some((DeviceBuffer::upload(slice, stream)?, off))
                }
                None => None,
            };
            Ok(Some(Self { data_type: c.data_type(), len: c.len(), dbhip_type: t, is_scalar: false, data: DeviceBuffer::upload(bytes, stream)?, validity, size }))
        };
        let none = DecimalSize::default();
        use databend_common_expression::types::DecimalColumn as D;
        use databend_common_expression::types::NumberColumn as N;
        match inner {
            Column::Number(N::Int8(b)) => up(DBHIP_T_I8, bytes_of(b), none),
            Column::Number(N::Int16(b)) => up(DBHIP_T_I16, bytes_of(b), none),
            Column::Number(N::Int32(b)) => up(DBHIP_T_I32, bytes_of(b), none),
            Column::Number(N::Int64(b)) => up(DBHIP_T_I64, bytes_of(b), none),
            Column::Number(N::UInt8(b)) => up(DBHIP_T_U8, bytes_of(b), none),
            Column::Number(N::UInt16(b)) => up(DBHIP_T_U16, bytes_of(b), none),
            Column::Number(N::UInt32(b)) => up(DBHIP_T_U32, bytes_of(b), none),
            Column::Number(N::UInt64(b)) => up(DBHIP_T_U64, bytes_of(b), none),
            Column::Number(N::Float32(b)) => up(DBHIP_T_F32, bytes_of(b), none),
            Column::Number(N::Float64(b)) => up(DBHIP_T_F64, bytes_of(b), none),
            Column::Date(b) => up(DBHIP_T_DATE, bytes_of(b), none),
            Column::Timestamp(b) => up(DBHIP_T_TIMESTAMP, bytes_of(b), none),
            Column::Decimal(D::Decimal64(b, s)) => up(DBHIP_T_DEC64, bytes_of(b), *s),
            Column::Decimal(D::Decimal128(b, s)) => up(DBHIP_T_DEC128, bytes_of(b), *s),
            Column::Decimal(D::Decimal256(b, s)) => up(DBHIP_T_DEC256, bytes_of(b), *s),
            // Boolean: the Bitmap's bytes (LSB first) are the values; String: the 16-byte views + data buffers (binview/view.rs:30-42)
            // are handled by `strings.rs` of a full binding (dbhip_col.buffers); not part of this sketch's three traits.
            _ => Ok(None),
        }
    }

    fn from_scalar(s: &Scalar, ty: &DataType, n: usize, stream: *mut c_void) -> Result<Option<Self>> {
        // one-row column of the scalar's type, uploaded, is_scalar = 1
        let col = databend_common_expression::ColumnBuilder::repeat(&s.as_ref(), 1, ty).build();
        Ok(Self::from_column(&col, stream)?.map(|mut d| {
            d.is_scalar = true;
            d.len = n;
            d
        }))
    }
}

fn bytes_of<T>(b: &Buffer<T>) -> &[u8] {
    // Buffer<T> derefs to [T] (buffer/immutable.rs:62-74)
    unsafe { std::slice::from_raw_parts(b.as_ptr() as *const u8, b.len() * std::mem::size_of::<T>()) }
}

/// A device result wrapped back into a reference `Buffer<T>`: downloaded here; a device-resident pipeline would keep the
/// `DeviceBuffer` inside `Column::Opaque` and never copy.
pub fn buffer_from_device<T: Default + Clone>(d: &DeviceBuffer, n: usize, stream: *mut c_void) -> Result<Buffer<T>> {
    Ok(d.download::<T>(n, stream)?.into())
}

pub fn number_type_of(t: NumberDataType) -> i32 {
    match t {
        NumberDataType::Int8 => DBHIP_T_I8,
        NumberDataType::Int16 => DBHIP_T_I16,
        NumberDataType::Int32 => DBHIP_T_I32,
        NumberDataType::Int64 => DBHIP_T_I64,
        NumberDataType::UInt8 => DBHIP_T_U8,
        NumberDataType::UInt16 => DBHIP_T_U16,
        NumberDataType::UInt32 => DBHIP_T_U32,
        NumberDataType::UInt64 => DBHIP_T_U64,
        NumberDataType::Float32 => DBHIP_T_F32,
        NumberDataType::Float64 => DBHIP_T_F64,
    }
}
