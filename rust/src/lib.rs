//! `struct BLU` of the `blu` crate (rwl/blu 0.2.1, `src/blu.rs:9-335`) on an MI355X: the same public methods,
//! argument meaning and `Status` values, with all numerical work done by the HIP kernels behind the C ABI of
//! `include/blu_hip.h` (`libblu_hip.so`).  A caller of the reference switches by replacing `use blu::BLU`
//! with `use blu_hip::BLU`.
//!
//! What differs, and why:
//! * `BLU.lu` is a public field as in the reference (`blu.rs:10`), but `struct LU` is not a struct of arrays in host
//!   memory (all state of the factorization lives in HBM): it owns the device handle and has the reference's public
//!   parameter fields as setter / getter pairs (`blu.lu.set_droptol(..)`, `blu.lu.droptol()`; `src/lu/lu.rs:11-66`)
//!   and its getters (`src/lu/lu.rs:398-684`) under the same names.  `BLU.realloc_factor` is a plain public field
//!   (`blu.rs:18-20`), handed to the library before every call that may grow storage.
//! * `Status::Reallocate` never reaches the caller of the object API in the reference either
//!   (`blu.rs:105-115, 277-283, 324-330`); the library grows its device storage itself.
//! * Two more error values exist: `ErrorDevice` (a HIP call failed, `last_error()` has the text) and
//!   `ErrorOutOfMemory` (`BLU_ERROR_OUT_OF_MEMORY` of the reference's C ancestry).
//! * `new` takes the HIP device ordinal through `BLU::new_on(m, b_nz, device)`; `BLU::new(m, b_nz)` uses device 0.
//!
//! NOT COMPILED in this repository's environment (no Rust toolchain in the image); the identical
//! call sequence is tested through the Python mirror `blu_amd/blu.py`.
#![allow(clippy::too_many_arguments)]

pub mod maxvolume;
pub use maxvolume::maxvolume;

use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_int};

/// BLU integer type (`src/lib.rs:32`).
pub type LUInt = i64;

/// `enum Status` of the reference (`src/lib.rs:38-64`) plus the two device-side failures.
#[derive(PartialEq, Clone, Copy, Debug)]
pub enum Status {
    Reallocate,
    WarningSingularMatrix,
    ErrorInvalidCall,
    ErrorArgumentMissing,
    ErrorInvalidArgument,
    ErrorMaximumUpdates,
    ErrorSingularUpdate,
    ErrorOutOfMemory,
    ErrorDevice,
}

#[repr(C)]
pub struct BluHip {
    _private: [u8; 0],
}

// include/blu_hip.h -- every entry cites the reference interface it replaces there.
extern "C" {
    fn blu_hip_new(m: i64, b_nz: i64, device: c_int) -> *mut BluHip;
    fn blu_hip_free(h: *mut BluHip);
    fn blu_hip_set_param(h: *mut BluHip, key: c_int, value: c_double) -> c_int;
    fn blu_hip_get_param(h: *const BluHip, key: c_int) -> c_double;
    fn blu_hip_get_stat(h: *const BluHip, key: c_int) -> c_double;
    fn blu_hip_factorize(h: *mut BluHip, b_begin: *const u64, b_end: *const u64, b_i: *const u64, b_x: *const f64, b_i_len: u64) -> c_int;
    fn blu_hip_get_factors(
        h: *mut BluHip, rowperm: *mut i64, colperm: *mut i64, l_colptr: *mut i64, l_rowidx: *mut i64, l_value: *mut f64,
        u_colptr: *mut i64, u_rowidx: *mut i64, u_value: *mut f64,
    ) -> c_int;
    fn blu_hip_solve_dense(h: *mut BluHip, rhs: *const f64, lhs: *mut f64, trans: c_char) -> c_int;
    fn blu_hip_solve_sparse(
        h: *mut BluHip, nzrhs: i64, irhs: *const u64, xrhs: *const f64, p_nzlhs: *mut i64, ilhs: *mut i64, lhs: *mut f64, trans: c_char,
    ) -> c_int;
    fn blu_hip_solve_for_update(
        h: *mut BluHip, nzrhs: i64, irhs: *const u64, xrhs: *const f64, p_nzlhs: *mut i64, ilhs: *mut i64, lhs: *mut f64, trans: c_char,
    ) -> c_int;
    fn blu_hip_update(h: *mut BluHip, xtbl: c_double) -> c_int;
    fn blu_hip_last_error(h: *const BluHip) -> *const c_char;
}

fn status_of(code: c_int) -> Result<(), Status> {
    match code {
        0 => Ok(()),
        1 => Err(Status::Reallocate),
        2 => Err(Status::WarningSingularMatrix),
        -2 => Err(Status::ErrorInvalidCall),
        -3 => Err(Status::ErrorArgumentMissing),
        -4 => Err(Status::ErrorInvalidArgument),
        -5 => Err(Status::ErrorMaximumUpdates),
        -6 => Err(Status::ErrorSingularUpdate),
        -9 => Err(Status::ErrorOutOfMemory),
        _ => Err(Status::ErrorDevice),
    }
}

// enum blu_param / enum blu_stat of include/blu_hip.h
mod key {
    pub const DROPTOL: i32 = 0;
    pub const ABSTOL: i32 = 1;
    pub const RELTOL: i32 = 2;
    pub const NZBIAS: i32 = 3;
    pub const MAXSEARCH: i32 = 4;
    pub const PAD: i32 = 5;
    pub const STRETCH: i32 = 6;
    pub const COMPRESS_THRES: i32 = 7;
    pub const SPARSE_THRES: i32 = 8;
    pub const SEARCH_ROWS: i32 = 9;
    pub const REALLOC_FACTOR: i32 = 10;
    pub const M: i32 = 0;
    pub const NUPDATE: i32 = 1;
    pub const NFACTORIZE: i32 = 2;
    pub const L_NZ: i32 = 3;
    pub const U_NZ: i32 = 4;
    pub const MIN_PIVOT: i32 = 5;
    pub const MAX_PIVOT: i32 = 6;
    pub const CONDEST_L: i32 = 7;
    pub const CONDEST_U: i32 = 8;
    pub const NORM_L: i32 = 9;
    pub const NORM_U: i32 = 10;
    pub const NORMEST_L_INV: i32 = 11;
    pub const NORMEST_U_INV: i32 = 12;
    pub const ONENORM: i32 = 13;
    pub const INFNORM: i32 = 14;
    pub const RESIDUAL_TEST: i32 = 15;
    pub const MATRIX_NZ: i32 = 16;
    pub const RANK: i32 = 17;
    pub const BUMP_SIZE: i32 = 18;
    pub const BUMP_NZ: i32 = 19;
    pub const NSEARCH_PIVOT: i32 = 20;
    pub const NEXPAND: i32 = 21;
    pub const NGARBAGE: i32 = 22;
    pub const FACTOR_FLOPS: i32 = 23;
    pub const TIME_FACTORIZE: i32 = 24;
    pub const NFORREST: i32 = 35;
    pub const PIVOT_ERROR: i32 = 36;
    pub const UPDATE_COST: i32 = 124;
}

/// `struct LU` (`src/lu/lu.rs`) as a user of the object API sees it: parameters and getters.  Owns the device handle.
pub struct LU {
    h: *mut BluHip,
}

macro_rules! param {
    ($get:ident, $set:ident, $key:expr, $t:ty) => {
        pub fn $get(&self) -> $t {
            unsafe { blu_hip_get_param(self.h, $key) as $t }
        }
        pub fn $set(&mut self, v: $t) {
            unsafe {
                blu_hip_set_param(self.h, $key, v as f64);
            }
        }
    };
}
macro_rules! stat {
    ($name:ident, $key:expr, $t:ty) => {
        pub fn $name(&self) -> $t {
            unsafe { blu_hip_get_stat(self.h, $key) as $t }
        }
    };
}

impl LU {
    // public parameter fields, lu.rs:11-66 (defaults lu.rs:249-259)
    param!(droptol, set_droptol, key::DROPTOL, f64);
    param!(abstol, set_abstol, key::ABSTOL, f64);
    param!(reltol, set_reltol, key::RELTOL, f64);
    param!(maxsearch, set_maxsearch, key::MAXSEARCH, usize);
    param!(pad, set_pad, key::PAD, usize);
    param!(stretch, set_stretch, key::STRETCH, f64);
    param!(compress_thres, set_compress_thres, key::COMPRESS_THRES, f64);
    param!(sparse_thres, set_sparse_thres, key::SPARSE_THRES, f64);
    param!(search_rows, set_search_rows, key::SEARCH_ROWS, usize);
    /// `nzbias: Option<usize>` (lu.rs:33-41): `None` is passed as -1.
    pub fn nzbias(&self) -> Option<usize> {
        let v = unsafe { blu_hip_get_param(self.h, key::NZBIAS) };
        if v < 0.0 { None } else { Some(v as usize) }
    }
    pub fn set_nzbias(&mut self, v: Option<usize>) {
        unsafe {
            blu_hip_set_param(self.h, key::NZBIAS, v.map(|x| x as f64).unwrap_or(-1.0));
        }
    }
    // getters, lu.rs:398-684
    stat!(m, key::M, usize);
    stat!(nfactorize, key::NFACTORIZE, usize);
    stat!(l_nz, key::L_NZ, usize);
    stat!(u_nz, key::U_NZ, usize);
    stat!(min_pivot, key::MIN_PIVOT, f64);
    stat!(max_pivot, key::MAX_PIVOT, f64);
    stat!(condest_l, key::CONDEST_L, f64);
    stat!(condest_u, key::CONDEST_U, f64);
    stat!(norm_l, key::NORM_L, f64);
    stat!(norm_u, key::NORM_U, f64);
    stat!(normest_l_inv, key::NORMEST_L_INV, f64);
    stat!(normest_u_inv, key::NORMEST_U_INV, f64);
    stat!(onenorm, key::ONENORM, f64);
    stat!(infnorm, key::INFNORM, f64);
    stat!(residual_test, key::RESIDUAL_TEST, f64);
    stat!(matrix_nz, key::MATRIX_NZ, usize);
    stat!(rank, key::RANK, usize);
    stat!(bump_size, key::BUMP_SIZE, usize);
    stat!(bump_nz, key::BUMP_NZ, usize);
    stat!(nsearch_pivot, key::NSEARCH_PIVOT, usize);
    stat!(nexpand, key::NEXPAND, usize);
    stat!(ngarbage, key::NGARBAGE, usize);
    stat!(factor_flops, key::FACTOR_FLOPS, usize);
    stat!(time_factorize, key::TIME_FACTORIZE, f64);
    stat!(nforrest, key::NFORREST, usize);
    stat!(pivot_error, key::PIVOT_ERROR, f64);
    /// `LU::update_cost()` (lu.rs:324-326).
    stat!(update_cost, key::UPDATE_COST, f64);
    /// `nupdate: Option<usize>` (lu.rs:91): `None` while there is no valid factorization.
    pub fn nupdate(&self) -> Option<usize> {
        let v = unsafe { blu_hip_get_stat(self.h, key::NUPDATE) };
        if v < 0.0 { None } else { Some(v as usize) }
    }
}

impl Drop for LU {
    fn drop(&mut self) {
        unsafe { blu_hip_free(self.h) }
    }
}

/// `struct BLU` (`src/blu.rs:9-20`).
pub struct BLU {
    /// `pub lu: LU` (`blu.rs:10`).
    pub lu: LU,
    m: usize,
    /// Solution of the last `solve_sparse` / `solve_for_update` (dense, `m` entries; `blu.rs:12`).
    pub lhs: Vec<f64>,
    /// Its pattern in the reference's order (`blu.rs:14`).
    pub ilhs: Vec<LUInt>,
    /// Number of nonzeros in `lhs` (`blu.rs:16`).
    pub nzlhs: usize,
    /// Arrays are reallocated for max(realloc_factor, 1.0) times the required size (`blu.rs:18-20`, default 1.5).
    pub realloc_factor: f64,
}

// The reference's BLU is `Send` (plain owned data); a handle is used by one thread at a time here as well.
unsafe impl Send for BLU {}

impl BLU {
    /// `BLU::new(m, b_nz)` (`blu.rs:61-91`) on HIP device 0.  Panics if no gfx950 device is usable: there is
    /// no CPU fallback.
    pub fn new(m: usize, b_nz: usize) -> BLU {
        Self::new_on(m, b_nz, 0).expect("blu_hip_new failed: no gfx950 device, bad argument or out of memory")
    }

    pub fn new_on(m: usize, b_nz: usize, device: i32) -> Option<BLU> {
        let h = unsafe { blu_hip_new(m as i64, b_nz as i64, device) };
        if h.is_null() {
            return None;
        }
        Some(BLU { lu: LU { h }, m, lhs: vec![0.0; m], ilhs: vec![0; m], nzlhs: 0, realloc_factor: 1.5 })
    }

    // the library grows its device storage itself (the loops of blu.rs:105-115, 277-283, 324-330): it gets the factor
    fn push_realloc_factor(&mut self) {
        unsafe {
            blu_hip_set_param(self.lu.h, key::REALLOC_FACTOR, self.realloc_factor);
        }
    }

    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(blu_hip_last_error(self.lu.h)).to_string_lossy().into_owned() }
    }

    /// `BLU::factorize` (`blu.rs:95-118`): column j of B is `b_i[b_begin[j]..b_end[j]]`, `b_x[..]`.
    pub fn factorize(&mut self, b_begin: &[usize], b_end: &[usize], b_i: &[usize], b_x: &[f64]) -> Result<(), Status> {
        if b_begin.len() < self.m || b_end.len() < self.m || b_i.len() != b_x.len() {
            return Err(Status::ErrorInvalidArgument);
        }
        self.push_realloc_factor();
        // usize == u64 on the targets this back end exists for (x86-64 Linux hosts of MI355X nodes)
        let code = unsafe {
            blu_hip_factorize(
                self.lu.h, b_begin.as_ptr() as *const u64, b_end.as_ptr() as *const u64, b_i.as_ptr() as *const u64, b_x.as_ptr(), b_i.len() as u64,
            )
        };
        status_of(code)
    }

    /// `BLU::get_factors` (`blu.rs:139-160`, `get_factors.rs:48-180`): any `None` skips that output.
    pub fn get_factors(
        &mut self,
        rowperm: Option<&mut [LUInt]>, colperm: Option<&mut [LUInt]>,
        l_colptr: Option<&mut [LUInt]>, l_rowidx: Option<&mut [LUInt]>, l_value: Option<&mut [f64]>,
        u_colptr: Option<&mut [LUInt]>, u_rowidx: Option<&mut [LUInt]>, u_value: Option<&mut [f64]>,
    ) -> Result<(), Status> {
        fn p<T>(o: Option<&mut [T]>) -> *mut T {
            o.map(|s| s.as_mut_ptr()).unwrap_or(std::ptr::null_mut())
        }
        let code = unsafe {
            blu_hip_get_factors(self.lu.h, p(rowperm), p(colperm), p(l_colptr), p(l_rowidx), p(l_value), p(u_colptr), p(u_rowidx), p(u_value))
        };
        status_of(code)
    }

    /// `BLU::solve_dense` (`blu.rs:182-184`).
    pub fn solve_dense(&mut self, rhs: &[f64], lhs: &mut [f64], trans: char) -> Result<(), Status> {
        if rhs.len() < self.m || lhs.len() < self.m {
            return Err(Status::ErrorInvalidArgument);
        }
        status_of(unsafe { blu_hip_solve_dense(self.lu.h, rhs.as_ptr(), lhs.as_mut_ptr(), trans as c_char) })
    }

    // lu_clear_lhs, blu.rs:380-395
    fn clear_lhs(&mut self) {
        let nzsparse = (self.lu.sparse_thres() * self.m as f64) as usize;
        if self.nzlhs != 0 {
            if self.nzlhs <= nzsparse {
                for p in 0..self.nzlhs {
                    self.lhs[self.ilhs[p] as usize] = 0.0;
                }
            } else {
                self.lhs.iter_mut().for_each(|x| *x = 0.0);
            }
            self.nzlhs = 0;
        }
    }

    /// `BLU::solve_sparse` (`blu.rs:207-225`): the solution is left in `self.lhs` / `self.ilhs[..self.nzlhs]`.
    pub fn solve_sparse(&mut self, nzrhs: LUInt, irhs: &[usize], xrhs: &[f64], trans: char) -> Result<(), Status> {
        // the C side reads irhs[0..nzrhs) and xrhs[0..nzrhs): never hand it a count beyond the slices
        if nzrhs < 0 || nzrhs as usize > irhs.len() || nzrhs as usize > xrhs.len() {
            return Err(Status::ErrorInvalidArgument);
        }
        self.clear_lhs();
        let mut nz: i64 = 0;
        let code = unsafe {
            blu_hip_solve_sparse(
                self.lu.h, nzrhs, irhs.as_ptr() as *const u64, xrhs.as_ptr(), &mut nz, self.ilhs.as_mut_ptr(), self.lhs.as_mut_ptr(), trans as c_char,
            )
        };
        status_of(code)?;
        self.nzlhs = nz as usize;
        Ok(())
    }

    /// `BLU::solve_for_update` (`blu.rs:257-288`).  With `want_solution != 0` the solution is left in
    /// `self.lhs` / `self.ilhs[..self.nzlhs]`; otherwise only the update is prepared.
    pub fn solve_for_update(&mut self, nzrhs: usize, irhs: &[usize], xrhs: Option<&[f64]>, trans: char, want_solution: LUInt) -> Result<(), Status> {
        // the C side reads irhs[0..nzrhs) / xrhs[0..nzrhs), and irhs[0] for a transposed solve even when nzrhs == 0
        let transposed = trans == 't' || trans == 'T';
        if nzrhs > irhs.len() || xrhs.map_or(false, |x| nzrhs > x.len()) || (transposed && irhs.is_empty()) {
            return Err(Status::ErrorInvalidArgument);
        }
        self.clear_lhs();
        self.push_realloc_factor();
        let xp = xrhs.map(|x| x.as_ptr()).unwrap_or(std::ptr::null());
        let mut nz: i64 = 0;
        let code = unsafe {
            if want_solution != 0 {
                blu_hip_solve_for_update(
                    self.lu.h, nzrhs as i64, irhs.as_ptr() as *const u64, xp, &mut nz, self.ilhs.as_mut_ptr(), self.lhs.as_mut_ptr(), trans as c_char,
                )
            } else {
                blu_hip_solve_for_update(
                    self.lu.h, nzrhs as i64, irhs.as_ptr() as *const u64, xp, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(),
                    trans as c_char,
                )
            }
        };
        status_of(code)?;
        if want_solution != 0 {
            self.nzlhs = nz as usize;
        }
        Ok(())
    }

    /// `BLU::update` (`blu.rs:319-335`).
    pub fn update(&mut self, xtbl: f64) -> Result<(), Status> {
        self.push_realloc_factor();
        status_of(unsafe { blu_hip_update(self.lu.h, xtbl) })
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    // examples/simple.rs:20-33 of the reference: x_i = 0.1 (i + 1).  Needs an MI355X.
    #[test]
    fn simple_rs() {
        let arow: Vec<usize> = vec![0, 7, 8, 1, 4, 9, 2, 9, 3, 6, 7, 8, 9, 1, 4, 5, 3, 6, 9, 0, 3, 7, 8, 0, 3, 7, 8, 1, 2, 3, 6, 9];
        let acolst: Vec<usize> = vec![0, 3, 6, 8, 13, 15, 16, 19, 23, 27, 32];
        let a = vec![
            2.1, 0.14, 0.09, 1.1, 0.06, 0.03, 1.7, 0.04, 1.0, 0.32, 0.19, 0.32, 0.44, 0.06, 1.6, 2.2, 0.32, 1.9, 0.43, 0.14, 0.19, 1.1, 0.22, 0.09,
            0.32, 0.22, 2.4, 0.03, 0.04, 0.44, 0.43, 3.2,
        ];
        let b = vec![0.403, 0.28, 0.55, 1.504, 0.812, 1.32, 1.888, 1.168, 2.473, 3.695];
        let mut blu = BLU::new(10, 32);
        blu.factorize(&acolst[..10], &acolst[1..], &arow, &a).unwrap();
        let mut x = vec![0.0; 10];
        blu.solve_dense(&b, &mut x, 'N').unwrap();
        for (i, xi) in x.iter().enumerate() {
            assert!((xi - 0.1 * (i as f64 + 1.0)).abs() < 1e-13);
        }
    }
}
