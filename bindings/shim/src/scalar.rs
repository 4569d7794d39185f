//! `impl ScalarFunction` (expression/src/function.rs:103-105) over the element-wise kernels. Registered BEFORE the CPU overloads so
//! that the candidate order (function.rs:275-340: overload id = registration order) prefers the device implementation; a status
//! outside OK / ROW_ERRORS hands the call to the CPU closure the registry already holds (`fallback`).
use std::ffi::c_void;
use std::ptr::null_mut;

use databend_common_expression::types::AnyType;
use databend_common_expression::types::DataType;
use databend_common_expression::types::NumberDataType;
use databend_common_expression::Column;
use databend_common_expression::EvalContext;
use databend_common_expression::ScalarFunction;
use databend_common_expression::Value;

use crate::device::number_type_of;
use crate::device::DeviceBuffer;
use crate::device::DeviceColumn;
use crate::sys::*;

/// plus / minus / multiply / divide / div / modulo over number columns (numeric_basic_arithmetic.rs:255-544) -> `dbhip_arith`.
pub struct HipArith {
    pub op: i32,                       // dbhip_arith_op
    pub args: [DataType; 2],
    pub out: NumberDataType,           // ResultTypeOfBinary, computed by the registry's own type rules (dbhip_arith_result_type agrees)
    pub fallback: Box<dyn ScalarFunction>,
}

impl ScalarFunction for HipArith {
    fn eval(&self, args: &[Value<AnyType>], ctx: &mut EvalContext) -> Value<AnyType> {
        let n = ctx.num_rows;
        let stream: *mut c_void = null_mut();
        let run = || -> databend_common_exception::Result<Option<(Value<AnyType>, Vec<u8>, u64)>> {
            let (Some(l), Some(r)) = (
                DeviceColumn::from_value(&args[0], &self.args[0], n, stream)?,
                DeviceColumn::from_value(&args[1], &self.args[1], n, stream)?,
            ) else {
                return Ok(None);
            };
            let out_t = number_type_of(self.out);
            let es = self.out.bit_width() as usize / 8;
            let out = DeviceBuffer::alloc(n * es)?;
            // the error Bitmap starts all ones; a row that raises has its bit CLEARED (EvalContext::set_error, function.rs:534-556)
            let err = DeviceBuffer::upload(&vec![0xFFu8; (n + 63) / 64 * 8], stream)?;
            let cnt = DeviceBuffer::upload(&[0u64], stream)?;
            let (lc, rc) = (l.as_col(), r.as_col());
            let st = unsafe { dbhip_arith(self.op, &lc, &rc, n as i64, out_t, out.ptr(), err.ptr() as *mut u8, cnt.ptr() as *mut u64, stream) };
            if st != DBHIP_OK && st != DBHIP_ERR_ROW_ERRORS {
                return Ok(None); // DBHIP_ERR_UNSUPPORTED (a type pair the kernels do not take): the CPU overload
            }
            let nerr = cnt.download::<u64>(1, stream)?[0];
            let bits = if nerr > 0 { err.download::<u8>((n + 7) / 8, stream)? } else { vec![] };
            let col = number_column_from_device(self.out, &out, n, stream)?;
            Ok(Some((Value::Column(col), bits, nerr)))
        };
        match run() {
            Ok(Some((v, bits, nerr))) => {
                if nerr > 0 {
                    // first failing row, as render_error reports it (function.rs:567-620); NULL rows never raise (ctx.validity)
                    if let Some(row) = (0..n).find(|i| bits[i >> 3] >> (i & 7) & 1 == 0) {
                        ctx.set_error(row, "divided by zero");
                    }
                }
                v
            }
            _ => self.fallback.eval(args, ctx),
        }
    }
}

/// eq / noteq / lt / lte / gt / gte (comparison.rs:98-112) -> `dbhip_cmp`: the result is a Boolean column = a Bitmap.
pub struct HipCmp {
    pub op: i32, // dbhip_cmp_op
    pub args: [DataType; 2],
    pub fallback: Box<dyn ScalarFunction>,
}

impl ScalarFunction for HipCmp {
    fn eval(&self, args: &[Value<AnyType>], ctx: &mut EvalContext) -> Value<AnyType> {
        let n = ctx.num_rows;
        let stream: *mut c_void = null_mut();
        let run = || -> databend_common_exception::Result<Option<Value<AnyType>>> {
            let (Some(l), Some(r)) = (
                DeviceColumn::from_value(&args[0], &self.args[0], n, stream)?,
                DeviceColumn::from_value(&args[1], &self.args[1], n, stream)?,
            ) else {
                return Ok(None);
            };
            let words = (n + 63) / 64;
            let out = DeviceBuffer::alloc(words * 8)?;
            let (lc, rc) = (l.as_col(), r.as_col());
            let st = unsafe { dbhip_cmp(self.op, &lc, &rc, n as i64, out.ptr() as *mut u8, stream) };
            if st != DBHIP_OK {
                return Ok(None);
            }
            let bytes = out.download::<u8>(words * 8, stream)?;
            Ok(Some(Value::Column(Column::Boolean(databend_common_column::bitmap::Bitmap::from_u8_vec(bytes, n)))))
        };
        match run() {
            Ok(Some(v)) => v,
            _ => self.fallback.eval(args, ctx),
        }
    }
}

fn number_column_from_device(t: NumberDataType, d: &DeviceBuffer, n: usize, stream: *mut c_void) -> databend_common_exception::Result<Column> {
    use databend_common_expression::types::NumberColumn as N;
    use crate::device::buffer_from_device as b;
    Ok(Column::Number(match t {
        NumberDataType::Int8 => N::Int8(b(d, n, stream)?),
        NumberDataType::Int16 => N::Int16(b(d, n, stream)?),
        NumberDataType::Int32 => N::Int32(b(d, n, stream)?),
        NumberDataType::Int64 => N::Int64(b(d, n, stream)?),
        NumberDataType::UInt8 => N::UInt8(b(d, n, stream)?),
        NumberDataType::UInt16 => N::UInt16(b(d, n, stream)?),
        NumberDataType::UInt32 => N::UInt32(b(d, n, stream)?),
        NumberDataType::UInt64 => N::UInt64(b(d, n, stream)?),
        NumberDataType::Float32 => N::Float32(b::<f32>(d, n, stream)?.into_iter().map(Into::into).collect()),
        NumberDataType::Float64 => N::Float64(b::<f64>(d, n, stream)?.into_iter().map(Into::into).collect()),
    }))
}

// Registration (function.rs:87-134,372-379), for every (op, left, right) the kernels take:
//
//   registry.register_function(Function {
//       signature: FunctionSignature { name: "plus".into(), args_type: vec![l.clone(), r.clone()], return_type: out.clone() },
//       eval: FunctionEval::Scalar { calc_domain: cpu.calc_domain, derive_stat: None,
//                                    eval: Box::new(HipArith { op: DBHIP_OP_PLUS, args: [l, r], out: out_num, fallback: cpu.eval }) },
//   });
//
// `cpu` is the overload the registry already built for the same signature (its domain calculation is reused: domains are a
// planner-side matter, function.rs:99-101).
