//! examples/css_decoding.rs of the `ldpc` crate with the GPU decoder in the place of `BpDecoder`
//! (the only line that differs from the reference's example is the body of `decoder`), followed by the
//! same experiment as ONE fused device Monte-Carlo run (`GpuCssMonteCarlo`).
//! Not compiled in this repository's environment (no Rust toolchain there).
use ldpc::codes::{CssCode, LinearCode};
use ldpc::decoders::{CssDecoder, SyndromeDecoder};
use ldpc::noise::{DepolarizingNoise, Probability};
use ldpc_hip::{GpuBpDecoder, GpuCssMonteCarlo};
use pauli::PauliOperator;
use rand::thread_rng;

fn main() {
    let max_iterations = 100;
    let probability = Probability::new(0.01);
    let code = code();
    let decoder = decoder(&code, probability, max_iterations);
    let noise = DepolarizingNoise::with_probability(probability);
    let mut failures = 0;
    for _ in 0..1000 {
        let error = code.random_error(&noise, &mut thread_rng());
        let syndrome = code.syndrome_of(&error);
        let correction: PauliOperator = decoder.correction_for(syndrome.as_view()).into();
        if !code.has_stabilizer(&(&error * &correction)) {
            failures += 1;
        }
    }
    println!("{}", failures);

    // the whole loop on the device: noise keyed by (seed, shot), counters back
    let counters = GpuCssMonteCarlo::new(&code, probability, max_iterations).run(probability, 2024, 0, 10_000_000);
    println!("{} failures in {} shots", counters[1], counters[0]);
}

fn code() -> CssCode {
    let code = LinearCode::random_regular_code()
        .num_bits(20)
        .num_checks(15)
        .bit_degree(3)
        .check_degree(4)
        .sample_with(&mut thread_rng())
        .unwrap();
    CssCode::hypergraph_product(&code, &code)
}

fn decoder(code: &CssCode, probability: Probability, max_iterations: usize) -> CssDecoder<GpuBpDecoder> {
    CssDecoder {
        x: GpuBpDecoder::new(&code.stabilizers.x, probability, max_iterations),
        z: GpuBpDecoder::new(&code.stabilizers.z, probability, max_iterations),
    }
}
