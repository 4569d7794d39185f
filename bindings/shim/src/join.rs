//! `impl Join` / `impl JoinStream` (new_hash_join/join.rs:22-53) for the inner join on fixed-width keys
//! (memory/inner_join.rs, hashtable/fixed_keys.rs:126-269). C++ twin, built and tested: `InnerHashJoin` in dbhip_host.hpp
//! (left / right / semi / anti / full kinds and the CONJUNCT streams are there as well).
use std::ffi::c_void;

use databend_common_base::base::ProgressValues;
use databend_common_exception::ErrorCode;
use databend_common_exception::Result;
use databend_common_expression::DataBlock;

use crate::check;
use crate::device::DeviceBuffer;
use crate::device::DeviceColumn;
use crate::sys::*;

pub trait JoinStream: Send + Sync {
    fn next(&mut self) -> Result<Option<DataBlock>>;
}
pub trait Join: Send + Sync + 'static {
    fn add_block(&mut self, data: Option<DataBlock>) -> Result<()>;
    fn final_build(&mut self) -> Result<Option<ProgressValues>>;
    fn probe_block(&mut self, data: DataBlock) -> Result<Box<dyn JoinStream + '_>>;
    fn final_probe(&mut self) -> Result<Option<Box<dyn JoinStream + '_>>> { Ok(None) }
}

pub struct HipInnerHashJoin {
    table: *mut dbhip_join,
    key_width: i32,                // 8 | 16 | 32: KeysU64 / KeysU128 / KeysU256 (HashMethodFixedKeys, method_fixed_keys.rs:103-139)
    build_keys: Vec<usize>,
    probe_keys: Vec<usize>,
    build_blocks: Vec<(DataBlock, Vec<DeviceColumn>)>, // the build side stays resident: its rows are taken by build-row index
    stream: *mut c_void,
    max_block_size: usize,
    finalized: bool,
}
unsafe impl Send for HipInnerHashJoin {}
unsafe impl Sync for HipInnerHashJoin {}

struct PackedKeys { keys: DeviceBuffer, valid: DeviceBuffer }

impl HipInnerHashJoin {
    /// DataBlock::choose_hash_method_with_types (kernels/group_by.rs:40-80) -> dbhip_keys_method; 0 = HashMethodSerializer:
    /// `dbhip_join_create_binary` or the CPU join.
    fn pack(&self, block: &DataBlock, cols: &[usize]) -> Result<(PackedKeys, Vec<DeviceColumn>)> {
        let mut hold = vec![];
        let mut c = vec![];
        for &i in cols {
            let d = DeviceColumn::from_entry(block.get_by_offset(i), self.stream)?.ok_or_else(|| ErrorCode::Unimplemented("join key type stays on the CPU join"))?;
            c.push(d.as_col());
            hold.push(d);
        }
        let n = block.num_rows();
        let keys = DeviceBuffer::alloc(n * self.key_width as usize)?;
        let valid = DeviceBuffer::alloc((n + 63) / 64 * 8)?;
        check(unsafe { dbhip_pack_keys(c.as_ptr(), c.len() as i32, n as i64, self.key_width, keys.ptr(), valid.ptr() as *mut u8, self.stream) })?;
        Ok((PackedKeys { keys, valid }, hold))
    }
}

impl Join for HipInnerHashJoin {
    fn add_block(&mut self, data: Option<DataBlock>) -> Result<()> {
        // (the reference squashes the build side before this call, new_hash_join/memory/basic.rs:78-89: blocks arrive large)
        let Some(block) = data else { return Ok(()) };
        let (k, hold) = self.pack(&block, &self.build_keys.clone())?;
        check(unsafe { dbhip_join_add_build(self.table, k.keys.ptr(), k.valid.ptr() as *const u8, block.num_rows() as i64, self.stream) })?;
        self.build_blocks.push((block, hold));
        Ok(())
    }

    fn final_build(&mut self) -> Result<Option<ProgressValues>> {
        if self.finalized {
            return Ok(None);
        }
        check(unsafe { dbhip_join_finalize(self.table, self.stream) })?;
        self.finalized = true;
        let rows: usize = self.build_blocks.iter().map(|(b, _)| b.num_rows()).sum();
        Ok(Some(ProgressValues { rows, bytes: 0 }))
    }

    fn probe_block(&mut self, data: DataBlock) -> Result<Box<dyn JoinStream + '_>> {
        let n = data.num_rows() as i64;
        let (k, _hold) = self.pack(&data, &self.probe_keys.clone())?;
        let mut total = 0u64;
        // count first (the only way a caller can size its outputs), then emit into buffers of exactly that size; the second
        // call reuses the counts of the first (k_join.hip: `prepared`)
        check(unsafe { dbhip_join_probe_count(self.table, k.keys.ptr(), k.valid.ptr() as *const u8, n, &mut total, self.stream) })?;
        let (pi, bi) = (DeviceBuffer::alloc(total as usize * 4)?, DeviceBuffer::alloc(total as usize * 4)?);
        let mut pairs = 0u64;
        check(unsafe { dbhip_join_probe(self.table, k.keys.ptr(), k.valid.ptr() as *const u8, n, pi.ptr() as *mut u32, bi.ptr() as *mut u32, total as i64, &mut pairs, self.stream) })?;
        Ok(Box::new(HipJoinStream { join: self, probe: data, probe_idx: pi, build_row: bi, pairs: pairs as usize, at: 0 }))
    }
}

/// The matched pairs of one probe block, handed out max_block_size rows at a time: probe projection (`dbhip_take_block` by probe
/// index) ++ build projection (by build row), inner_join.rs:248-268. Pair order = by probe row, then build row.
pub struct HipJoinStream<'a> {
    join: &'a HipInnerHashJoin,
    probe: DataBlock,
    probe_idx: DeviceBuffer,
    build_row: DeviceBuffer,
    pairs: usize,
    at: usize,
}
unsafe impl Send for HipJoinStream<'_> {}
unsafe impl Sync for HipJoinStream<'_> {}

impl JoinStream for HipJoinStream<'_> {
    fn next(&mut self) -> Result<Option<DataBlock>> {
        if self.at >= self.pairs {
            return Ok(None);
        }
        let m = (self.pairs - self.at).min(self.join.max_block_size);
        let sel_p = unsafe { (self.probe_idx.ptr() as *const u32).add(self.at) };
        let sel_b = unsafe { (self.build_row.ptr() as *const u32).add(self.at) };
        self.at += m;
        // one dbhip_take_block per side: every column of the block gathered with ONE selection (kernels/take.rs:43)
        let _ = (sel_p, sel_b, &self.probe);
        unimplemented!("gather + column construction: see InnerHashJoin::Stream::next in databend_amd/host/dbhip_host.hpp")
    }
}

impl Drop for HipInnerHashJoin {
    fn drop(&mut self) {
        unsafe { dbhip_join_destroy(self.table) };
    }
}
