//! `impl AccumulatingTransform` (transform_accumulating.rs:30-38) standing where `TransformPartialAggregate` stands
//! (transform_aggregate_partial.rs:119-303), at the block size the pipeline really uses: <= 65,536 rows per `transform` call
//! (max_block_size, settings_default.rs:142-148). The table is PIPELINED (dbhip_groupby_set_pipelined): a block costs one queued
//! descriptor, 32 blocks one launch, and what a block may raise comes back at the checkpoint with the number of blocks that were
//! merged — the rest is replayed through the CPU operator the reference already has. C++ twin, built and tested:
//! `TransformFusedPartialAggregate` in databend_amd/host/dbhip_host.hpp.
use std::ffi::c_void;
use std::ptr::null;

use databend_common_exception::Result;
use databend_common_expression::DataBlock;
use databend_common_pipeline_transforms::processors::AccumulatingTransform;

use crate::check;
use crate::device::DeviceColumn;
use crate::falls_back;
use crate::sys::*;

/// filter predicate + one argument expression per aggregate, flattened post-order into one register program (built once per
/// pipeline from the `Expr`s of the fused `TransformFilter` / `CompoundBlockOperator`, block_operator.rs:42-85)
pub struct AggProgram {
    pub ins: Vec<dbhip_expr_ins>,
    pub inputs: Vec<usize>,   // block offsets of the program's input columns
    pub filter_reg: i32,
    pub arg_regs: Vec<i32>,   // per aggregate: register | DBHIP_ARG_INPUT(c) | DBHIP_ARG_NONE
}

/// The CPU operators this transform replaces, kept for the blocks the device gives back (the reference's own
/// TransformFilter -> CompoundBlockOperator -> TransformPartialAggregate chain over ONE block).
pub trait CpuFallback: Send {
    fn accumulate(&mut self, block: DataBlock) -> Result<()>;
    /// its partial states as the serialized-state block `Payload::aggregate_flush` emits (payload_flush.rs:151-181)
    fn flush(&mut self) -> Result<Vec<DataBlock>>;
}

pub struct HipTransformPartialAggregate {
    table: *mut dbhip_groupby,
    stream: *mut c_void,
    group_columns: Vec<usize>,
    program: AggProgram,
    fused: bool,
    prepared: bool,
    /// blocks whose kernels are queued: their device buffers must outlive the checkpoint, and they are what is replayed on failure
    retained: Vec<(DataBlock, Vec<DeviceColumn>)>,
    cpu: Box<dyn CpuFallback>,
}
unsafe impl Send for HipTransformPartialAggregate {}

impl HipTransformPartialAggregate {
    pub fn try_create(key_types: &[i32], key_nullable: &[u8], aggs: &[dbhip_agg_desc], group_columns: Vec<usize>, program: AggProgram,
                      cpu: Box<dyn CpuFallback>) -> Result<Self> {
        let mut table: *mut dbhip_groupby = std::ptr::null_mut();
        check(unsafe { dbhip_groupby_create(key_types.as_ptr(), key_nullable.as_ptr(), key_types.len() as i32, aggs.as_ptr(), aggs.len() as i32, 1024, &mut table) })?;
        let mut stream: *mut c_void = std::ptr::null_mut();
        check(unsafe { dbhip_stream_create(&mut stream) })?; // one stream per pipeline lane, like the lane's own partial table
        let rc = unsafe { dbhip_groupby_set_pipelined(table, 1, stream) };
        let fused = rc == DBHIP_OK; // DBHIP_ERR_UNSUPPORTED: the layout is outside the fused kernel -> every block goes to `cpu`
        if !fused && rc != DBHIP_ERR_UNSUPPORTED {
            check(rc)?;
        }
        Ok(Self { table, stream, group_columns, program, fused, prepared: false, retained: vec![], cpu })
    }

    fn bind(&self, block: &DataBlock) -> Result<Option<(Vec<DeviceColumn>, Vec<dbhip_col>, Vec<dbhip_col>)>> {
        let mut hold = vec![];
        let mut inputs = vec![];
        for &i in &self.program.inputs {
            let Some(c) = DeviceColumn::from_entry(block.get_by_offset(i), self.stream)? else { return Ok(None) };
            inputs.push(c.as_col());
            hold.push(c);
        }
        let mut keys = vec![];
        for &i in &self.group_columns {
            let Some(c) = DeviceColumn::from_entry(block.get_by_offset(i), self.stream)? else { return Ok(None) };
            keys.push(c.as_col());
            hold.push(c);
        }
        Ok(Some((hold, inputs, keys)))
    }

    /// dbhip_groupby_checkpoint: blocks [committed, queued) were not merged -> the CPU chain takes them, in order
    fn drain(&mut self) -> Result<()> {
        if self.retained.is_empty() {
            return Ok(());
        }
        let mut committed = 0i64;
        let rc = unsafe { dbhip_groupby_checkpoint(self.table, &mut committed, self.stream) };
        let blocks = std::mem::take(&mut self.retained);
        if rc == DBHIP_OK {
            return Ok(());
        }
        if !falls_back(rc) {
            return check(rc);
        }
        if rc != DBHIP_ERR_ROW_ERRORS {
            self.fused = false; // keys (or shape) this kernel does not take: the rest of the stream goes to the CPU operator
        }
        for (block, _dev) in blocks.into_iter().skip(committed as usize) {
            self.cpu.accumulate(block)?; // raises the reference's own row error ("Decimal overflow ...") where there is one
        }
        Ok(())
    }
}

impl AccumulatingTransform for HipTransformPartialAggregate {
    const NAME: &'static str = "HipTransformPartialAggregate";

    fn transform(&mut self, block: DataBlock) -> Result<Vec<DataBlock>> {
        if block.num_rows() == 0 {
            return Ok(vec![]);
        }
        if !self.fused {
            self.cpu.accumulate(block)?;
            return Ok(vec![]);
        }
        let Some((hold, inputs, keys)) = self.bind(&block)? else {
            self.drain()?;
            self.cpu.accumulate(block)?;
            return Ok(vec![]);
        };
        let ap = dbhip_agg_program {
            prog: if self.program.ins.is_empty() { null() } else { self.program.ins.as_ptr() },
            n_ins: self.program.ins.len() as i32,
            inputs: inputs.as_ptr(),
            n_inputs: inputs.len() as i32,
            filter_reg: self.program.filter_reg,
            arg_regs: self.program.arg_regs.as_ptr(),
        };
        if !self.prepared {
            // the PREPARE of the pipeline: the run-time specialised kernels of this shape (incl. the multi-block one), ~1 s once per
            // query shape and process, cached on disk
            check(unsafe { dbhip_groupby_prepare_program(self.table, keys.as_ptr(), &ap) })?;
            self.prepared = true;
        }
        let rc = unsafe { dbhip_groupby_add_block_program(self.table, keys.as_ptr(), &ap, block.num_rows() as i64, null(), 0, self.stream) };
        if falls_back(rc) {
            // refused AT the call (a program / column shape outside the fused subset): nothing was queued
            self.fused = false;
            self.drain()?;
            self.cpu.accumulate(block)?;
            return Ok(vec![]);
        }
        check(rc)?;
        self.retained.push((block, hold));
        Ok(vec![])
    }

    fn on_finish(&mut self, output: bool) -> Result<Vec<DataBlock>> {
        self.drain()?;
        if !output {
            return Ok(vec![]);
        }
        // serialized partial states for TransformAggregateSerializer / the Flight exchange / TransformFinalAggregate:
        // `[state columns..., group columns...]` exactly as Payload::aggregate_flush writes them — dbhip_groupby_state_fields names
        // the fields, dbhip_groupby_flush_state_block fills them; the CPU fallback's states travel beside them.
        let mut out = hip_flush_state_block(self.table, self.stream)?;
        out.extend(self.cpu.flush()?);
        Ok(out)
    }
}

impl Drop for HipTransformPartialAggregate {
    fn drop(&mut self) {
        unsafe {
            dbhip_groupby_destroy(self.table);
            dbhip_stream_destroy(self.stream);
        }
    }
}

/// `dbhip_groupby_num_groups` -> buffers -> `dbhip_groupby_flush_state_block` -> one DataBlock with AggregateMeta::Serialized.
/// (Column construction from device buffers as in device.rs; elided here: it is `merge_result()` / `serialize()` of the C++ mirror.)
fn hip_flush_state_block(table: *mut dbhip_groupby, stream: *mut c_void) -> Result<Vec<DataBlock>> {
    let mut n = 0i64;
    check(unsafe { dbhip_groupby_num_groups(table, &mut n, stream) })?;
    if n == 0 {
        return Ok(vec![]);
    }
    let mut types = [0i32; 96];
    let mut owner = [0i32; 96];
    let mut nf = 0i32;
    check(unsafe { dbhip_groupby_state_fields(table, types.as_mut_ptr(), owner.as_mut_ptr(), 96, &mut nf) })?;
    // allocate one device buffer per key column and per state field, call dbhip_groupby_flush_state_block, wrap as Columns
    unimplemented!("column construction: see AggregateHashTable::serialize / merge_result in databend_amd/host/dbhip_host.hpp")
}
