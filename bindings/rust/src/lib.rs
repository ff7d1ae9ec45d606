//! GPU decoders for the `ldpc` crate on MI355X, over the C ABI of include/ldpc_hip.h.
//!
//! `GpuBpDecoder` implements `LinearDecoder`, `SyndromeDecoder<SparseBinSlice, SparseBinVec>`
//! and `ClassicalSyndromeDecoder` (ldpc/src/decoders/mod.rs:15-23), so it drops into
//! `CssDecoder { x, z }` exactly where `BpDecoder` sits in examples/css_decoding.rs:41-44.
//! `GpuFlipDecoder<Code>::decode<T>` mirrors `FlipDecoder<Code>::decode<T>` (ldpc/src/decoders/flip.rs:17-33),
//! `GpuBinarySymmetricChannel`, `GpuDepolarizingNoise` and `GpuErasureChannel` implement `NoiseModel`
//! (ldpc/src/noise/mod.rs:24-30): the caller's `R: Rng` supplies the 64-bit seed of the device stream.
//!
//! This file is not compiled in the repository's environment (no Rust toolchain there).  Its `extern "C"` block is
//! checked against include/ldpc_hip.h, name by name and type by type, by tests/test_rust_shim_abi.py.
use ldpc::codes::LinearCode;
use ldpc::decoders::{ClassicalSyndromeDecoder, LinearDecoder, SyndromeDecoder};
use ldpc::noise::{NoiseModel, Probability};
use pauli::{Pauli, PauliOperator, X, Y, Z};
use rand::Rng;
use sparse_bin_mat::{SparseBinMat, SparseBinSlice, SparseBinVec, SparseBinVecBase};
use std::borrow::Borrow;
use std::ffi::CStr;
use std::fmt;
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
    fn ldpc_code_has_codeword(
        code: *const ldpc_code,
        len: usize,
        positions: *const usize,
        n_positions: usize,
        out: *mut c_int,
    ) -> c_int;
    fn ldpc_bsc_sample_error_of_length(
        probability: c_double,
        len: usize,
        seed: u64,
        shot: u64,
        out_positions: *mut usize,
        out_n: *mut usize,
    ) -> c_int;
    fn ldpc_depolarizing_sample_error_of_length(
        probability: c_double,
        len: usize,
        seed: u64,
        shot: u64,
        out_positions: *mut usize,
        out_paulis: *mut u8,
        out_n: *mut usize,
    ) -> c_int;
    fn ldpc_erasure_sample_error_of_length(
        probability: c_double,
        len: usize,
        seed: u64,
        shot: u64,
        out_positions: *mut usize,
        out_n: *mut usize,
    ) -> c_int;
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

struct Code_(*mut ldpc_code);

impl Code_ {
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
        Code_(handle)
    }
}

impl Drop for Code_ {
    fn drop(&mut self) {
        unsafe { ldpc_code_destroy(self.0) }
    }
}

/// Drop-in for `ldpc::decoders::BpDecoder` (belief_propagation.rs:8-12).
pub struct GpuBpDecoder {
    handle: *mut ldpc_bp_decoder,
    code: Code_,
    num_bits: usize,
    num_checks: usize,
}

// Handles serialize calls internally and hold no thread-affine state.
unsafe impl Send for GpuBpDecoder {}
unsafe impl Sync for GpuBpDecoder {}

impl GpuBpDecoder {
    /// `BpDecoder::new(&parity_mat, probability, num_iterations)` (belief_propagation.rs:36-46)
    pub fn new(parity_mat: &SparseBinMat, probability: Probability, num_iterations: usize) -> Self {
        let code = Code_::from_matrix(parity_mat);
        let mut handle = std::ptr::null_mut();
        check(unsafe { ldpc_bp_create(code.0, probability.value(), num_iterations, &mut handle) });
        Self {
            handle,
            code,
            num_bits: parity_mat.number_of_columns(),
            num_checks: parity_mat.number_of_rows(),
        }
    }

    /// `BpDecoder::num_bits` (belief_propagation.rs:78-80)
    pub fn num_bits(&self) -> usize {
        self.num_bits
    }

    /// `BpDecoder::num_checks` (belief_propagation.rs:82-84)
    pub fn num_checks(&self) -> usize {
        self.num_checks
    }

    /// `BpDecoder::has_zero_syndrome` (belief_propagation.rs:86-88): `(H * vector).is_zero()`
    pub fn has_zero_syndrome(&self, vector: SparseBinSlice) -> bool {
        let positions = vector.as_slice();
        let mut out: c_int = 0;
        check(unsafe {
            ldpc_code_has_codeword(self.code.0, vector.len(), positions.as_ptr(), positions.len(), &mut out)
        });
        out != 0
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

/// Drop-in for `ldpc::decoders::FlipDecoder<Code>` (flip.rs:7-9): generic over how the code is held
/// (`LinearCode`, `&LinearCode`, `Rc<LinearCode>`, ...: `Code: Borrow<LinearCode>`, flip.rs:17-19).
pub struct GpuFlipDecoder<Code> {
    code: Code,
    handle: *mut ldpc_flip_decoder,
    num_bits: usize,
}

unsafe impl<Code: Send> Send for GpuFlipDecoder<Code> {}
unsafe impl<Code: Sync> Sync for GpuFlipDecoder<Code> {}

impl<Code> GpuFlipDecoder<Code>
where
    Code: Borrow<LinearCode>,
{
    /// `FlipDecoder::new(code)` (flip.rs:12-14)
    pub fn new(code: Code) -> Self {
        let (handle, num_bits) = {
            let linear: &LinearCode = code.borrow();
            let c = Code_::from_matrix(linear.parity_check_matrix());
            let mut handle = std::ptr::null_mut();
            check(unsafe { ldpc_flip_create(c.0, &mut handle) });
            (handle, linear.len())
        };
        Self { code, handle, num_bits }
    }

    /// `FlipDecoder::decode(&message)` (flip.rs:21-33), for owned vectors and views alike
    pub fn decode<T>(&self, message: &SparseBinVecBase<T>) -> SparseBinVec
    where
        T: std::ops::Deref<Target = [usize]>,
    {
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

    /// The code the decoder was built for (flip.rs:45-47 keeps it private; exposed here for callers of `mc_run`).
    pub fn code(&self) -> &LinearCode {
        self.code.borrow()
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

/// flip.rs:50-54
impl<T> fmt::Display for GpuFlipDecoder<T> {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "Flip decoder")
    }
}

impl<Code> Drop for GpuFlipDecoder<Code> {
    fn drop(&mut self) {
        unsafe { ldpc_flip_destroy(self.handle) }
    }
}

// ---- noise models (ldpc/src/noise) ------------------------------------------------------------
// The device streams are keyed by (seed, shot) (noise specification).  Behind the crate's
// `NoiseModel::sample_error_of_length<R: Rng>(&self, length, &mut R)` the caller's generator supplies
// the seed: one `u64` drawn per sample, shot index 0.  The distributions are the reference's
// (i.i.d. Bernoulli(p) positions; a uniform non-trivial Pauli on each); the streams are this library's.

/// Drop-in for `ldpc::noise::BinarySymmetricChannel` (binary_symmetric_channel.rs:15-44).
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct GpuBinarySymmetricChannel {
    probability: f64,
}

impl GpuBinarySymmetricChannel {
    /// binary_symmetric_channel.rs:22-29
    pub fn with_probability(probability: Probability) -> Self {
        Self { probability: probability.value() }
    }
}

impl NoiseModel for GpuBinarySymmetricChannel {
    type Error = SparseBinVec;

    /// binary_symmetric_channel.rs:35-43
    fn sample_error_of_length<R: Rng>(&self, length: usize, rng: &mut R) -> Self::Error {
        let mut out = vec![0usize; length];
        let mut n = 0usize;
        check(unsafe {
            ldpc_bsc_sample_error_of_length(self.probability, length, rng.gen::<u64>(), 0, out.as_mut_ptr(), &mut n)
        });
        out.truncate(n);
        SparseBinVec::new(length, out)
    }
}

/// binary_symmetric_channel.rs:46-50
impl fmt::Display for GpuBinarySymmetricChannel {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "BSC({})", self.probability)
    }
}

/// Drop-in for `ldpc::noise::DepolarizingNoise` (depolarizing.rs:12-49).
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct GpuDepolarizingNoise {
    probability: f64,
}

impl GpuDepolarizingNoise {
    /// depolarizing.rs:20-28
    pub fn with_probability(probability: Probability) -> Self {
        Self { probability: probability.value() }
    }
}

impl NoiseModel for GpuDepolarizingNoise {
    type Error = PauliOperator;

    /// depolarizing.rs:34-48
    fn sample_error_of_length<R: Rng>(&self, length: usize, rng: &mut R) -> Self::Error {
        let mut positions = vec![0usize; length];
        let mut codes = vec![0u8; length];
        let mut n = 0usize;
        check(unsafe {
            ldpc_depolarizing_sample_error_of_length(
                self.probability,
                length,
                rng.gen::<u64>(),
                0,
                positions.as_mut_ptr(),
                codes.as_mut_ptr(),
                &mut n,
            )
        });
        positions.truncate(n);
        // include/ldpc_hip.h: 1 = X, 2 = Y, 3 = Z
        let paulis: Vec<Pauli> = codes[..n].iter().map(|c| match c { 1 => X, 2 => Y, _ => Z }).collect();
        PauliOperator::new(length, positions, paulis)
    }
}

/// depolarizing.rs:51-55
impl fmt::Display for GpuDepolarizingNoise {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "Depolarizing Noise (prob = {})", self.probability)
    }
}

/// Drop-in for `ldpc::noise::ErasureChannel` (erasure.rs:14-43).
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct GpuErasureChannel {
    probability: f64,
}

impl GpuErasureChannel {
    /// erasure.rs:21-28
    pub fn with_probability(probability: Probability) -> Self {
        Self { probability: probability.value() }
    }
}

impl NoiseModel for GpuErasureChannel {
    type Error = SparseBinVec;

    /// erasure.rs:34-42
    fn sample_error_of_length<R: Rng>(&self, length: usize, rng: &mut R) -> Self::Error {
        let mut out = vec![0usize; length];
        let mut n = 0usize;
        check(unsafe {
            ldpc_erasure_sample_error_of_length(self.probability, length, rng.gen::<u64>(), 0, out.as_mut_ptr(), &mut n)
        });
        out.truncate(n);
        SparseBinVec::new(length, out)
    }
}

/// erasure.rs:45-49
impl fmt::Display for GpuErasureChannel {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "Erasure({})", self.probability)
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
        let hx = Code_::from_matrix(code.x_stabs_binary());
        let hz = Code_::from_matrix(code.z_stabs_binary());
        let lx = Code_::from_matrix(code.x_logicals_binary());
        let lz = Code_::from_matrix(code.z_logicals_binary());
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
