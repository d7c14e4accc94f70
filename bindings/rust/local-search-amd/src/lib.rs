//! Safe wrapper over `libls_amd.so` for the reference's plug-in seam:
//! `GpuContext::nqueens_best_of_neighbourhood` is what replaces lines 315-325 of
//! `LocalSearch::execute` (local-search/src/local_search.rs) for the N-queens plug-in.
//! SOURCE ONLY — not compiled in this repository's environment (no Rust toolchain).
pub mod sys;

use std::ffi::CStr;

pub struct GpuContext { raw: *mut sys::ls_ctx }

// one context is driven by one thread at a time (the reference's Solution/Score are Send, not Sync)
unsafe impl Send for GpuContext {}

#[derive(Debug)]
pub struct LsError { pub code: i32, pub message: String }

pub struct NeighbourhoodBest { pub n_scored: usize, pub best: Option<(i64, usize, i64)> } // (score, column, new row)

impl GpuContext {
    pub fn new(device: i32) -> Result<Self, LsError> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::ls_ctx_create(device, &mut raw) };
        if rc != sys::LS_OK {
            let msg = unsafe { CStr::from_ptr(sys::ls_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(LsError { code: rc, message: msg });
        }
        Ok(Self { raw })
    }

    fn check(&self, rc: i32) -> Result<(), LsError> {
        if rc == sys::LS_OK { return Ok(()); }
        let msg = unsafe { CStr::from_ptr(sys::ls_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(LsError { code: rc, message: msg })
    }

    /// get_col_scores + sum (examples/nqueens/src/lib.rs:74-87,130-139).
    pub fn nqueens_col_scores(&mut self, rows: &[i64]) -> Result<(Vec<i64>, i64), LsError> {
        let packed: Vec<i32> = rows.iter().map(|&r| r as i32).collect();
        let mut cs = vec![0i64; rows.len()];
        let mut total = 0i64;
        let rc = unsafe { sys::ls_nqueens_col_scores(self.raw, 1, rows.len() as i32, packed.as_ptr(), cs.as_mut_ptr(), &mut total) };
        self.check(rc)?;
        Ok((cs, total))
    }

    /// iter_local_moves(cols order).filter(!tabu).map(score).take(window).sort().first()
    pub fn nqueens_best_of_neighbourhood(&mut self, rows: &[i64], cur_score: i64, cols: &[usize], window: usize)
        -> Result<NeighbourhoodBest, LsError> {
        let packed: Vec<i32> = rows.iter().map(|&r| r as i32).collect();
        let cols32: Vec<i32> = cols.iter().map(|&c| c as i32).collect();
        let mut out = sys::ls_nq_result::default();
        let rc = unsafe { sys::ls_nqueens_eval(self.raw, rows.len() as i32, packed.as_ptr(), cur_score, cols32.as_ptr(),
                                               cols32.len() as i32, window as i64, &mut out, std::ptr::null_mut()) };
        self.check(rc)?;
        let best = (out.best_index >= 0).then(|| (out.best_score, out.best_col as usize, out.best_value as i64));
        Ok(NeighbourhoodBest { n_scored: out.n_scored as usize, best })
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) { unsafe { sys::ls_ctx_destroy(self.raw); } }
}
