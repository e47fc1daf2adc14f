// Golden-vector dump for the MI355X engine's oracle (see README.md in this directory).  NOT part of the reference: a new binary
// that calls the reference's public loaders / forwards on deterministic probe inputs and writes the outputs as .npy files.
//   cargo run --release --bin pin -- <probe_dir> <out_dir>
use std::env;
use std::fs::{self, File};
use std::io::Write;

use stablediffusion::backend::Backend;
use stablediffusion::model::autoencoder::{load::load_decoder, load::load_encoder, Decoder, Encoder};
use stablediffusion::model::unet::{load::load_unet, UNet};

use burn::tensor::Tensor;
use burn_tch::{LibTorch, LibTorchDevice};

// the reference's probe input recipe (src/bin/test/main.rs:51-54): sin(arange(n)) reshaped
fn arb_tensor<B: Backend, const D: usize>(dims: [usize; D], device: &B::Device) -> Tensor<B, D> {
    let n: usize = dims.iter().product();
    Tensor::arange(0..n as i64, device).float().sin().reshape(dims)
}

// minimal .npy v1.0 writer: little-endian f32, C order
fn write_npy(path: &str, shape: &[usize], data: &[f32]) {
    let dims = shape.iter().map(|d| d.to_string()).collect::<Vec<_>>().join(", ");
    let tuple = if shape.len() == 1 { format!("({},)", dims) } else { format!("({})", dims) };
    let mut header = format!("{{'descr': '<f4', 'fortran_order': False, 'shape': {}, }}", tuple);
    let unpadded = 10 + header.len() + 1;
    header.push_str(&" ".repeat((64 - unpadded % 64) % 64));
    header.push('\n');
    let mut f = File::create(path).expect("create npy");
    f.write_all(b"\x93NUMPY\x01\x00").unwrap();
    f.write_all(&(header.len() as u16).to_le_bytes()).unwrap();
    f.write_all(header.as_bytes()).unwrap();
    for v in data {
        f.write_all(&v.to_le_bytes()).unwrap();
    }
}

fn dump<B: Backend, const D: usize>(name: &str, out_dir: &str, t: Tensor<B, D>) {
    let shape = t.dims().to_vec();
    let data: Vec<f32> = t.into_data().convert::<f32>().value;
    write_npy(&format!("{}/{}.npy", out_dir, name), &shape, &data);
    println!("wrote {} {:?}", name, shape);
}

fn main() {
    type B = LibTorch<f32>; // or burn_ndarray::NdArray<f32> with NdArrayDevice::Cpu
    let device = LibTorchDevice::Cpu;
    let args: Vec<String> = env::args().collect();
    let (probe, out) = (args[1].clone(), args[2].clone());
    fs::create_dir_all(&out).unwrap();

    // tiny UNet (oracle/config.py tiny_config: adm 8, context 20): shapes of src/bin/test/main.rs:133-136 on an 8x8 latent
    let unet: UNet<B> = load_unet(&format!("{}/params_unet", probe), &device).unwrap();
    let x = arb_tensor::<B, 4>([1, 4, 8, 8], &device);
    let context = arb_tensor::<B, 3>([1, 1, 20], &device);
    let y = arb_tensor::<B, 2>([1, 8], &device);
    let t = Tensor::<B, 1, burn::tensor::Int>::from_ints([1], &device);
    dump("unet_out", &out, unet.forward(x, t, context, y));

    // tiny VAE (tiny_vae_config): src/bin/test/main.rs:147,158
    let encoder: Encoder<B> = load_encoder(&format!("{}/params_vae/encoder", probe), &device).unwrap();
    dump("encoder_out", &out, encoder.forward(arb_tensor::<B, 4>([1, 3, 16, 16], &device)));
    let decoder: Decoder<B> = load_decoder(&format!("{}/params_vae/decoder", probe), &device).unwrap();
    dump("decoder_out", &out, decoder.forward(arb_tensor::<B, 4>([1, 4, 4, 4], &device)));
}
