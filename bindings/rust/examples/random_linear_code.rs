//! examples/random_linear_code.rs of the `ldpc` crate, then the same code decoded on the GPU:
//! one word through `GpuFlipDecoder::decode`, and a device Monte-Carlo run of 10^6 shots.
//! Not compiled in this repository's environment (no Rust toolchain there).
use ldpc::codes::LinearCode;
use ldpc::noise::{BinarySymmetricChannel, Probability};
use ldpc_hip::GpuFlipDecoder;
use rand::thread_rng;

fn main() {
    let code = LinearCode::random_regular_code()
        .num_bits(4)
        .num_checks(3)
        .bit_degree(3)
        .check_degree(4)
        .sample_with(&mut thread_rng())
        .unwrap();

    let probability = Probability::new(0.2);
    let noise = BinarySymmetricChannel::with_probability(probability);
    let error = code.random_error(&noise, &mut thread_rng());
    println!("{}", error);

    let decoder = GpuFlipDecoder::new(&code);
    println!("{}", decoder.decode(&error));
    let counters = decoder.mc_run(probability, 2024, 0, 1_000_000);
    println!("{} failures in {} shots", counters[1], counters[0]);
}
