//! GPU decoders for the `ldpc` crate on MI355X, over the C ABI of include/ldpc_hip.h.
//!
//! `GpuBpDecoder` implements `LinearDecoder`, `SyndromeDecoder<SparseBinSlice, SparseBinVec>`
//! and `ClassicalSyndromeDecoder` (ldpc/src/decoders/mod.rs:15-23), so it drops into
//! `CssDecoder { x, z }` exactly where `BpDecoder` sits in examples/css_decoding.rs:41-44.
//! `GpuFlipDecoder::decode` mirrors `FlipDecoder::decode` (ldpc/src/decoders/flip.rs:21-33).
//!
//! This file is not compiled in the repository's environment (no Rust toolchain there).
use ldpc::decoders::{ClassicalSyndromeDecoder, LinearDecoder, SyndromeDecoder};
use ldpc::noise::Probability;
use sparse_bin_mat::{SparseBinMat, SparseBinSlice, SparseBinVec};
use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_int};

#[repr(C)]
pub struct ldpc_code {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ldpc_flip_decoder {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ldpc_bp_decoder {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ldpc_css_code {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ldpc_css_bp_decoder {
    _private: [u8; 0],
}

pub const LDPC_COUNTERS: usize = 8;

extern "C" {
    fn ldpc_last_error() -> *const c_char;
    fn ldpc_code_create(
        n_bits: usize,
        n_checks: usize,
        row_ptr: *const usize,
        col_idx: *const usize,
        out: *mut *mut ldpc_code,
    ) -> c_int;
    fn ldpc_code_destroy(code: *mut ldpc_code);
    fn ldpc_flip_create(code: *const ldpc_code, out: *mut *mut ldpc_flip_decoder) -> c_int;
    fn ldpc_flip_decode(
        dec: *mut ldpc_flip_decoder,
        len: usize,
        positions: *const usize,
        n_positions: usize,
        out_positions: *mut usize,
        out_n: *mut usize,
    ) -> c_int;
    fn ldpc_flip_mc_run(
        dec: *mut ldpc_flip_decoder,
        probability: c_double,
        seed: u64,
        first_shot: u64,
        n_shots: usize,
        counters: *mut u64,
        d_errors_out: *mut u64,
        d_syndromes_out: *mut u64,
        d_words_out: *mut u64,
    ) -> c_int;
    fn ldpc_flip_destroy(dec: *mut ldpc_flip_decoder);
    fn ldpc_bp_create(
        code: *const ldpc_code,
        probability: c_double,
        max_iterations: usize,
        out: *mut *mut ldpc_bp_decoder,
    ) -> c_int;
    fn ldpc_bp_correction_for(
        dec: *mut ldpc_bp_decoder,
        syndrome_len: usize,
        syndrome_positions: *const usize,
        n_positions: usize,
        out_positions: *mut usize,
        out_n: *mut usize,
    ) -> c_int;
    fn ldpc_bp_decode(
        dec: *mut ldpc_bp_decoder,
        len: usize,
        positions: *const usize,
        n_positions: usize,
        out_positions: *mut usize,
        out_n: *mut usize,
    ) -> c_int;
    fn ldpc_bp_mc_run(
        dec: *mut ldpc_bp_decoder,
        noise_probability: c_double,
        seed: u64,
        first_shot: u64,
        n_shots: usize,
        counters: *mut u64,
        d_failed_out: *mut u8,
    ) -> c_int;
    fn ldpc_bp_destroy(dec: *mut ldpc_bp_decoder);
    fn ldpc_css_create_with_logicals(
        x_stabs: *const ldpc_code,
        z_stabs: *const ldpc_code,
        x_logicals: *const ldpc_code,
        z_logicals: *const ldpc_code,
        out: *mut *mut ldpc_css_code,
    ) -> c_int;
    fn ldpc_css_destroy(css: *mut ldpc_css_code);
    fn ldpc_css_bp_create(
        css: *const ldpc_css_code,
        probability: c_double,
        max_iterations: usize,
        out: *mut *mut ldpc_css_bp_decoder,
    ) -> c_int;
    fn ldpc_css_bp_mc_run(
        dec: *mut ldpc_css_bp_decoder,
        noise_probability: c_double,
        seed: u64,
        first_shot: u64,
        n_shots: usize,
        counters: *mut u64,
    ) -> c_int;
    fn ldpc_css_bp_destroy(dec: *mut ldpc_css_bp_decoder);
}

/// Status codes other than 0 become the panics the reference raises for the same
/// misuse (length mismatch: linear_code/mod.rs:397-403; bad probability: noise/mod.rs:36-37).
fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { CStr::from_ptr(ldpc_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

struct Code(*mut ldpc_code);

impl Code {
    fn from_matrix(matrix: &SparseBinMat) -> Self {
        let mut row_ptr = vec![0usize];
        let mut col_idx = Vec::with_capacity(matrix.number_of_ones());
        for row in matrix.rows() {
            col_idx.extend(row.non_trivial_positions());
            row_ptr.push(col_idx.len());
        }
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            ldpc_code_create(
                matrix.number_of_columns(),
                matrix.number_of_rows(),
                row_ptr.as_ptr(),
                col_idx.as_ptr(),
                &mut handle,
            )
        });
        Code(handle)
    }
}

impl Drop for Code {
    fn drop(&mut self) {
        unsafe { ldpc_code_destroy(self.0) }
    }
}

/// Drop-in for `ldpc::decoders::BpDecoder` (belief_propagation.rs:8-12).
pub struct GpuBpDecoder {
    handle: *mut ldpc_bp_decoder,
    num_bits: usize,
    num_checks: usize,
}

// Handles serialize calls internally and hold no thread-affine state.
unsafe impl Send for GpuBpDecoder {}
unsafe impl Sync for GpuBpDecoder {}

impl GpuBpDecoder {
    /// `BpDecoder::new(&parity_mat, probability, num_iterations)` (belief_propagation.rs:36-46)
    pub fn new(parity_mat: &SparseBinMat, probability: Probability, num_iterations: usize) -> Self {
        let code = Code::from_matrix(parity_mat);
        let mut handle = std::ptr::null_mut();
        check(unsafe { ldpc_bp_create(code.0, probability.value(), num_iterations, &mut handle) });
        Self {
            handle,
            num_bits: parity_mat.number_of_columns(),
            num_checks: parity_mat.number_of_rows(),
        }
    }

    /// Whole Monte-Carlo run on device: BSC(noise) -> syndrome -> BP -> failure test.
    pub fn mc_run(&self, noise: Probability, seed: u64, first_shot: u64, n_shots: usize) -> [u64; LDPC_COUNTERS] {
        let mut counters = [0u64; LDPC_COUNTERS];
        check(unsafe {
            ldpc_bp_mc_run(
                self.handle,
                noise.value(),
                seed,
                first_shot,
                n_shots,
                counters.as_mut_ptr(),
                std::ptr::null_mut(),
            )
        });
        counters
    }
}

impl Drop for GpuBpDecoder {
    fn drop(&mut self) {
        unsafe { ldpc_bp_destroy(self.handle) }
    }
}

impl<'a> SyndromeDecoder<SparseBinSlice<'a>, SparseBinVec> for GpuBpDecoder {
    fn correction_for(&self, syndrome: SparseBinSlice) -> SparseBinVec {
        let mut out = vec![0usize; self.num_bits];
        let mut n = 0usize;
        let positions = syndrome.as_slice();
        check(unsafe {
            ldpc_bp_correction_for(
                self.handle,
                syndrome.len(),
                positions.as_ptr(),
                positions.len(),
                out.as_mut_ptr(),
                &mut n,
            )
        });
        out.truncate(n);
        SparseBinVec::new(self.num_bits, out)
    }
}

impl<'a> ClassicalSyndromeDecoder<'a> for GpuBpDecoder {}

impl LinearDecoder for GpuBpDecoder {
    fn decode(&self, message: SparseBinSlice) -> SparseBinVec {
        let mut out = vec![0usize; self.num_bits];
        let mut n = 0usize;
        let positions = message.as_slice();
        check(unsafe {
            ldpc_bp_decode(
                self.handle,
                message.len(),
                positions.as_ptr(),
                positions.len(),
                out.as_mut_ptr(),
                &mut n,
            )
        });
        out.truncate(n);
        SparseBinVec::new(self.num_bits, out)
    }
}

/// Drop-in for `ldpc::decoders::FlipDecoder` (flip.rs:7-9).
pub struct GpuFlipDecoder {
    handle: *mut ldpc_flip_decoder,
    num_bits: usize,
}

unsafe impl Send for GpuFlipDecoder {}
unsafe impl Sync for GpuFlipDecoder {}

impl GpuFlipDecoder {
    /// `FlipDecoder::new(code)` (flip.rs:12-14)
    pub fn new(code: &ldpc::codes::LinearCode) -> Self {
        let c = Code::from_matrix(code.parity_check_matrix());
        let mut handle = std::ptr::null_mut();
        check(unsafe { ldpc_flip_create(c.0, &mut handle) });
        Self { handle, num_bits: code.len() }
    }

    /// `FlipDecoder::decode(&message)` (flip.rs:21-33)
    pub fn decode(&self, message: &SparseBinVec) -> SparseBinVec {
        let mut out = vec![0usize; self.num_bits];
        let mut n = 0usize;
        let positions = message.as_slice();
        check(unsafe {
            ldpc_flip_decode(
                self.handle,
                message.len(),
                positions.as_ptr(),
                positions.len(),
                out.as_mut_ptr(),
                &mut n,
            )
        });
        out.truncate(n);
        SparseBinVec::new(self.num_bits, out)
    }

    pub fn mc_run(&self, noise: Probability, seed: u64, first_shot: u64, n_shots: usize) -> [u64; LDPC_COUNTERS] {
        let mut counters = [0u64; LDPC_COUNTERS];
        check(unsafe {
            ldpc_flip_mc_run(
                self.handle,
                noise.value(),
                seed,
                first_shot,
                n_shots,
                counters.as_mut_ptr(),
                std::ptr::null_mut(),
                std::ptr::null_mut(),
                std::ptr::null_mut(),
            )
        });
        counters
    }
}

impl Drop for GpuFlipDecoder {
    fn drop(&mut self) {
        unsafe { ldpc_flip_destroy(self.handle) }
    }
}

/// The loop of examples/css_decoding.rs:14-21 (sample a depolarizing error, syndrome, decode X and Z
/// parts with BP, test `has_stabilizer`) as one device run.  The code's own stabilizers and logicals
/// are handed over as they are (`CssCode::{x,z}_stabs_binary`, `{x,z}_logicals_binary`), so the failure
/// test is the crate's.  counters: [shots, failures, detected, undetected, iterations, capped, xfail, zfail].
pub struct GpuCssMonteCarlo {
    css: *mut ldpc_css_code,
    handle: *mut ldpc_css_bp_decoder,
}

unsafe impl Send for GpuCssMonteCarlo {}
unsafe impl Sync for GpuCssMonteCarlo {}

impl GpuCssMonteCarlo {
    pub fn new(code: &ldpc::codes::CssCode, probability: Probability, num_iterations: usize) -> Self {
        let hx = Code::from_matrix(code.x_stabs_binary());
        let hz = Code::from_matrix(code.z_stabs_binary());
        let lx = Code::from_matrix(code.x_logicals_binary());
        let lz = Code::from_matrix(code.z_logicals_binary());
        let mut css = std::ptr::null_mut();
        check(unsafe { ldpc_css_create_with_logicals(hx.0, hz.0, lx.0, lz.0, &mut css) });
        let mut handle = std::ptr::null_mut();
        check(unsafe { ldpc_css_bp_create(css, probability.value(), num_iterations, &mut handle) });
        Self { css, handle }
    }

    pub fn run(&self, noise: Probability, seed: u64, first_shot: u64, n_shots: usize) -> [u64; LDPC_COUNTERS] {
        let mut counters = [0u64; LDPC_COUNTERS];
        check(unsafe {
            ldpc_css_bp_mc_run(self.handle, noise.value(), seed, first_shot, n_shots, counters.as_mut_ptr())
        });
        counters
    }
}

impl Drop for GpuCssMonteCarlo {
    fn drop(&mut self) {
        unsafe {
            ldpc_css_bp_destroy(self.handle);
            ldpc_css_destroy(self.css);
        }
    }
}
