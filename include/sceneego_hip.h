/*
 * sceneego_hip.h — C ABI of libsceneego_hip.so (gfx950 / MI355X).
 *
 * The SceneEgo reference (jianwang-mpi/SceneEgo) is pure Python and has no FFI / plugin registry:
 * its depth-aware voxel pose path dispatches ATen operators (and numpy for the voxeliser).  This
 * library is what stands in for those operator calls; each entry point below names the reference
 * call site (file:line, relative to the reference repo) it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; no torch / HIP types in the signatures (`stream` is a hipStream_t
 *     passed as void*, NULL = default stream);
 *   - every pointer is DEVICE memory owned by the caller; nothing is allocated, freed or
 *     synchronised inside; launches are asynchronous on `stream` and hipGraph-capturable;
 *   - return value: 0 on success, otherwise the hipError_t of the failed call, or
 *     SE_ERR_BAD_ARG (-1) for an unsupported shape/argument (nothing is launched then);
 *   - activations are float32, channels-last: volumes are [B][Z][Y][X][C] ("NDHWC"; the reference's
 *     meshgrid order i,j,k of voxel_net_depth.py:117-121 is our Z,Y,X, flat voxel n = (i*G + j)*G + k),
 *     images are [B][H][W][C].
 *
 * Parity surface: every *_f32 / *_f64 entry point computes in the reference's arithmetic type and is covered by the
 * parity tests.  NOT part of the parity surface (opt-in, reduced precision, never selected by se_conv3d_f32 and never
 * part of bench.py's `value`): the *_bf16 entry points (bfloat16 storage, BASELINE configs[2]) and the three
 * se_conv3d_split3_* / se_conv3d_k3_split3_f32 entry points (EXPERIMENTAL: float32 tensors, 16-bit-mantissa products).
 */
#ifndef SCENEEGO_HIP_H
#define SCENEEGO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SE_ERR_BAD_ARG (-1)

/* epilogue flags of se_conv3d_f32 / se_deconv3d_k2s2_f32 */
#define SE_EPI_RELU          1   /* y = max(y, 0)                                                    */
#define SE_EPI_RES_PRE_RELU  2   /* y += residual BEFORE the ReLU  (Res3DBlock: relu(res + skip))     */
#define SE_EPI_RES_POST_RELU 4   /* y += residual AFTER the ReLU   (decoder: upsample(x) + skip_x)    */
#define SE_EPI_OUT_PLANAR    8   /* write [B][Cout][voxels] (NCDHW) instead of NDHWC                  */
#define SE_IN_PLANAR3        16  /* se_conv3d_f32, k = 7, cout = 16, dim % 8 == 0, no residual only: `in` is
                                  * triplet-planar [B][ceil(cin/3)][D][D][D][3] (channel c at triplet c/3,
                                  * slot c%3; slots >= cin must be finite) - see se_unproject_gather_planar3_f32 */
#define SE_IN_OCTET          32  /* se_conv3d_f32, k = 3 shapes with se_conv3d_f32_algo() == 2 only: `in` is octet-planar          */
#define SE_OUT_OCTET         64  /* [B][C/8][D][D][D][8] (channel c at octet c/8, slot c%8) / `out` is written that way /          */
#define SE_RES_OCTET         128 /* the skip tensor `residual` is read that way                                                 */
#define SE_EPI_SKIPCONV16    256 /* set by se_conv3d_skip16_f32 only (not a caller flag of se_conv3d_f32)                       */
#define SE_IN_QUAD           1024 /* se_conv3d_f32, k = 3 launches with se_conv3d_f32_variant(..., these flags) == 3 only: `in` is QUAD-planar   */
#define SE_OUT_QUAD          2048 /* [B][C/4][D][D][D][4] (channel c at quad c/4, slot c%4) / `out` is written that way /                     */
#define SE_RES_QUAD          4096 /* the skip tensor `residual` is read that way.  Quad- and octet-planar flags do not mix in one launch.    */
#define SE_LAYOUT_OCTET_BITS (SE_IN_OCTET | SE_OUT_OCTET | SE_RES_OCTET)
#define SE_LAYOUT_QUAD_BITS  (SE_IN_QUAD | SE_OUT_QUAD | SE_RES_QUAD)

/* ABI version; bumped on any signature change. */
int se_abi_version(void);

/* Occupancy voxeliser.  Replaces VoxelNetwork_depth.depth_map_to_voxel_numpy +
 * point_cloud_to_voxel_numpy (network/voxel_net_depth.py:194-222; per-sample host loop :252-255).
 *   depth    [B][depth_h][depth_w] float32 metres
 *   ray_tab  [up][up][3] float64: unit ray of padded-image pixel (x'+pad_x, y), indexed [y][x']
 *   occ      [B][G][G][G] float32, written as {0,1} (cleared inside)
 * Per pixel (y, x') of the `up` x `up` nearest-resized depth (src row floor(y*depth_h/up), src col
 * floor(x'*depth_w/up)): p = ray*d in float64; q = rint(((p.x+side/2)*G)/side, ((p.y+side/2)*G)/side,
 * (p.z*G)/side) (half-to-even, unfused); if 0<=q<=G-1 on all axes occ[qx][qy][qz] = 1.  If pad_x > 0
 * the zero-padded columns contribute the point (0,0,0).  Bit-exact with the reference.          */
int se_voxelize_f64(const float* depth, const double* ray_tab, float* occ,
                    int batch, int depth_h, int depth_w, int up, int pad_x,
                    int volume_size, double cuboid_side, void* stream);

/* Same, writing straight into the V2V input buffer buf [B][G^3][stride_c]: channels [c_offset, c_offset+4) of every voxel
 * are cleared, then channel c_offset receives the occupancy (replaces torch.stack/unsqueeze/cat,
 * network/voxel_net_depth.py:256-262, for the with_scene && !with_intersection case).                   */
int se_voxelize_strided_f64(const float* depth, const double* ray_tab, float* buf,
                            int batch, int depth_h, int depth_w, int up, int pad_x,
                            int volume_size, double cuboid_side, int stride_c, int c_offset, void* stream);

/* Same for a full-width depth map without resize/pad: dataset/real_depth_utils.py:29-60
 * (`voxel_output=True` path).  ray_tab [depth_h][depth_w][3] float64 indexed [y][x].            */
int se_voxelize_full_f64(const float* depth, const double* ray_tab, float* occ,
                         int batch, int depth_h, int depth_w,
                         int volume_size, double cuboid_side, void* stream);

/* Table-driven bilinear voxel gather.  Replaces nn.Upsample(1024^2) + ConstantPad2d(128) +
 * F.grid_sample (network/voxel_net_depth.py:60-61,238,243; utils/op.py:194-214) without the
 * 1024x1280 intermediate.
 *   feat [B][texels][channels] float32 (NHWC feature map, texels = H*W)
 *   idx  [voxels][4] int32 texel index per tap (-1 = zero tap), w [voxels][4] float32 tap weights
 *   out  [B][voxels][out_stride_c]; channels [out_c_offset, out_c_offset+channels) are written.
 * channels % 4 == 0, out_stride_c % 4 == 0, out_c_offset % 4 == 0.                                */
int se_unproject_gather_f32(const float* feat, const int* idx, const float* w, float* out,
                            int batch, int texels, int channels, int voxels,
                            int out_stride_c, int out_c_offset, void* stream);

/* with_intersection=True input assembly (network/voxel_net_depth.py:258-260): given vol channels
 * [0,c) already in `buf` [B][voxels][stride_c] and occ [B][voxels], writes buf[..., c:2c] = vol*occ. */
int se_intersection_f32(float* buf, const float* occ, int batch, int voxels, int channels,
                        int stride_c, void* stream);

/* Fused bias (+ residual) (+ ReLU) epilogue for the backbone's MIOpen 2D convolutions (BatchNorm folded into weight
 * and bias): replaces the bn / `out += residual` / relu element-wise passes of Bottleneck.forward
 * (network/pose_resnet.py:72-90).  x, residual (or NULL), out: [batch][channels][hw] float32, out may alias x. */
int se_bias_act_nchw_f32(const float* x, const float* bias, const float* residual, float* out,
                         int batch, int channels, int hw, int relu, void* stream);

/* Stem tail of the backbone in one pass: `bn1` (folded into conv1's weights: + bias), `relu` and `maxpool` (3x3, stride 2, padding 1),
 * network/pose_resnet.py:229-232, on the raw result of conv1: out = relu(max over the window (x) + bias[c]).
 * x [batch][channels][2 ho][2 wo], out [batch][channels][ho][wo] float32; wo % 4 == 0. */
int se_bias_relu_maxpool3x3s2_f32(const float* x, const float* bias, float* out, int batch, int channels, int ho, int wo, void* stream);

/* 1x1 convolution (stride 1) of the backbone as ONE float32 MFMA GEMM with its whole epilogue: replaces conv1 / conv3 / the stride-1
 * downsample of Bottleneck.forward with their BatchNorm (folded into w and bias), `out += residual` and the ReLU
 * (network/pose_resnet.py:72-90) - MIOpen's convolution plus the se_bias_act_nchw_f32 pass behind it (round 6; csrc/conv2d_1x1.hip).
 *   x [batch][cin][hw], residual (or NULL) / out [batch][cout][hw] float32 NCHW;  bias [cout]
 *   wpack = the folded [cout][cin] matrix as [cout / BC][cin / 16][BC][16] with BC = se_conv2d_1x1_tile_f32(batch, cin, cout, hw)
 *   (128 when cout % 128 == 0, else 64; 0 = shape not covered: cin % 16, cout % 64, hw % 16, batch * hw % 64 must be 0).
 *   in_bias (or NULL) [cin]: x is the RAW result of the producing convolution and its bias + ReLU are applied on the way in,
 *   x' = max(x + in_bias[c], 0) - conv2's `bn2` + `relu` (pose_resnet.py:79-81) folded into conv3's launch.
 * float32 in, float32 accumulate: differs from the MIOpen result by summation order only. */
int se_conv2d_1x1_tile_f32(int batch, int cin, int cout, int hw);
int se_conv2d_1x1_f32(const float* x, const float* wpack, const float* bias, const float* residual, const float* in_bias, float* out,
                      int batch, int cin, int cout, int hw, int relu, void* stream);

/* The same operator for the launches of batch 1-2 that would leave most CUs without a workgroup (64 - 1024 pixels against megabytes of
 * weights): 64 pixels x 16 channels per workgroup, the k steps over four / two wave groups.  wpack16 = [cout / 16][cin / 16][16][16];
 * cin % 64 == 0, cout % 16 == 0, hw % 16 == 0, batch * hw % 64 == 0 (SE_ERR_BAD_ARG otherwise).  Arguments as se_conv2d_1x1_f32. */
int se_conv2d_1x1_small_f32(const float* x, const float* wpack16, const float* bias, const float* residual, const float* in_bias, float* out,
                            int batch, int cin, int cout, int hw, int relu, void* stream);
/* ... stride 2: x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo]; the conditions above on (cin, cout, ho * wo), wo % 4 == 0. */
int se_conv2d_1x1_small_s2_f32(const float* x, const float* wpack16, const float* bias, float* out, int batch, int cin, int cout, int ho,
                               int wo, int relu, void* stream);

/* ... and its stride-2 form, the `downsample` convolution of a stage's first Bottleneck (network/pose_resnet.py:140-146):
 * x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo] = W x[:, :, ::2, ::2] + bias (+ ReLU); wpack / covered shapes as
 * se_conv2d_1x1_tile_f32(batch, cin, cout, ho * wo) says, wo % 4 == 0. */
int se_conv2d_1x1_s2_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int ho, int wo,
                         int relu, void* stream);

/* 3x3 convolution (stride 1, padding 1) of the backbone's deep stages as a direct float32 MFMA product: replaces conv2 (`conv3x3`,
 * network/pose_resnet.py:22-25) of the Bottlenecks of layer3 / layer4 with its folded BatchNorm (round 6; csrc/conv2d_3x3.hip).
 *   x [batch][cin][h][w], out [batch][cout][h][w] float32 NCHW; bias [cout] or NULL (the raw sums: the consumer adds it, see in_bias above)
 *   wpack = the folded [cout][cin][3][3] tensor as [cout / BC][cin / 16][9][BC][16] with BC = se_conv2d_3x3_tile_f32(batch, cin, cout, h, w)
 *   (32 or 16; 0 = shape not covered: cin % 32, cout % 32 must be 0 and the map 8k x 8m or 4k x 16m).
 * float32 in, float32 accumulate: differs from the MIOpen result by summation order only. */
int se_conv2d_3x3_tile_f32(int batch, int cin, int cout, int h, int w);
int se_conv2d_3x3_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int h, int w,
                      int relu, void* stream);

/* ... and its stride-2 form (conv2 of a stage's first Bottleneck): x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo], padding 1;
 * wpack = [cout / 16][cin / 16][9][16][16] (channel tile 16); cin % 32 == 0, cout % 16 == 0, output map 8k x 8m or 4k x 16m. */
int se_conv2d_3x3_s2_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int ho, int wo,
                         int relu, void* stream);

/* Output side of the 2-D pose head's transposed convolutions - ConvTranspose2d(k=4, s=2, p=1) + BatchNorm2d + ReLU,
 * network/pose_resnet.py:205-224 (built), :238 (run) - when the layer is computed as ONE GEMM over the un-shifted input:
 *   z    [batch][4 ky][4 kx][cout][h][w] = W_tap [cout x cin] @ x[b] [cin x h*w] for each of the 16 taps (any GEMM library;
 *        the host side uses rocBLAS through torch.matmul), BatchNorm scale folded into W
 *   out  [batch][cout][2h][2w] = bias[co] + the four taps that reach each output pixel (zero outside the map), ReLU if `relu`. */
int se_deconv2d_k4s2_assemble_f32(const float* z, const float* bias, float* out, int batch, int cout, int h, int w,
                                  int relu, void* stream);

/* Weight preparation: folds an eval-mode BatchNorm3d into the convolution and re-orders the weights
 * into the MFMA fragment order the conv kernels read (v_mfma_f32_16x16x4_f32 A-operand blocks).
 * Replaces nothing at run time in the reference — it is what makes Conv3d+BatchNorm3d(+ReLU)
 * (network/v2v.py:12-14,24-30,35-38,61-63) one kernel.
 *   w      Conv3d weight [cout][cin][k][k][k], or (transposed != 0) ConvTranspose3d weight [cin][cout][2][2][2]
 *   b      conv bias [cout] or NULL
 *   gamma,beta,mean,var  BatchNorm3d weight/bias/running_mean/running_var [cout], or all NULL (no BN)
 *   wpack  out, se_conv3d_packed_elems(...) floats;  bpack out, cout_pad floats (cout rounded up to 16)
 * cin_pad >= cin is the channel count of the activation tensor the conv will read (multiple of 16;
 * extra channels get zero weights).                                                               */
int se_conv3d_pack_f32(const float* w, const float* b, const float* gamma, const float* beta,
                       const float* mean, const float* var, float eps,
                       float* wpack, float* bpack,
                       int cout, int cin, int cin_pad, int ksize, int transposed, void* stream);
long long se_conv3d_packed_elems(int cout, int cin_pad, int ksize, int transposed);

/* Conv3d (k = 1, 3 or 7, stride 1, zero padding (k-1)/2) + folded BN + epilogue.
 * Replaces Basic3DBlock / Res3DBlock convs and output_layer (network/v2v.py:8-43,161).
 *   in  [B][D][D][D][cin_pad]   out [B][D][D][D][cout] (or planar, SE_EPI_OUT_PLANAR)
 *       (flags & SE_IN_PLANAR3, k = 7 only: in is [B][ceil(cin/3)][D][D][D][3]; cin_pad then only names the packed weights)
 *   residual: same shape as out (NDHWC) or NULL.  cin = real input channels (<= cin_pad, the channel stride
 *   of `in`; channels [cin, cin_pad) must hold finite values, they meet zero weights); cin_pad % 16 == 0;
 *   cout % 16 == 0 unless planar.  Volumes with dim % 8 == 0 and dim >= 16 run the LDS-tiled kernels, everything
 *   else the direct kernel; results are identical in both up to float32 summation order.
 *   workspace (optional, may be NULL): `workspace_elems` floats of scratch; when given, small volumes
 *   (too few voxels to fill 256 CUs) split the 27 taps over extra workgroups and reduce the partial sums
 *   from the workspace in a fixed order (deterministic), and the 7^3 front layer of a launch with fewer than two
 *   tiles per CU (batch 1 at 64^3) splits every tile's 3-channel chunks between two workgroups - the second halves'
 *   sums pass through the first B * D^3 * 16 floats of the workspace and are added by a second kernel of the same
 *   call.  32 Mi floats cover every V2V level up to B = 64.  A workspace serves one stream at a time. */
int se_conv3d_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                  float* out, int batch, int dim, int cin, int cin_pad, int cout, int ksize, int flags,
                  float* workspace, long long workspace_elems, void* stream);

/* se_conv3d_f32 that also writes max_pool3d(out, kernel 2, stride 2) from the kernel's epilogue: `pool_out` is channels-last
 * [B][D/2][D/2][D/2][cout] float32.  Stands in for a Res3DBlock's last convolution followed by encoder_pool (reference
 * network/v2v.py:104-119) without re-reading the block output.  Only shapes with se_conv3d_f32_algo() == 2 (the 2-D Winograd
 * kernel) and cin_pad == cin; SE_ERR_BAD_ARG otherwise.  `out` is written as by se_conv3d_f32 (either layout). */
int se_conv3d_pool_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                       float* out, float* pool_out, int batch, int dim, int cin, int cin_pad, int cout, int ksize, int flags,
                       float* workspace, long long workspace_elems, void* stream);

/* 3x3x3 convolution + folded BN with the Res3DBlock's 1x1x1 skip convolution computed in the same launch (reference
 * network/v2v.py:21-43: res_branch's second convolution and skip_con of a block whose channel count changes):
 *     out = act( conv3(in; wpack) + skip_w . skip_in + bpack )
 * `skip_in` is the block input, 16 channels, channels-last [B][D][D][D][16] - or, with SE_RES_QUAD beside SE_IN_QUAD | SE_OUT_QUAD,
 * quad-planar [B][4][D][D][D][4] (what se_conv3d_k7_fft_f32 writes with SE_OUT_QUAD); `skip_w` its BN-folded weights [cout][16];
 * `bpack` must hold the SUM of both folded biases.  2-D Winograd shapes (se_conv3d_f32_algo() == 2) with octet-planar `in`
 * and `out` only (flags must carry SE_IN_OCTET | SE_OUT_OCTET, or SE_IN_QUAD | SE_OUT_QUAD where se_conv3d_f32_variant(..., those
 * flags) == 3; SE_EPI_RELU optional); SE_ERR_BAD_ARG otherwise.  Saves the
 * 1x1x1 launch, its output tensor and the skip-tensor read of the 3x3x3 convolution. */
int se_conv3d_skip16_f32(const float* in, const float* wpack, const float* bpack, const float* skip_in, const float* skip_w,
                         float* out, int batch, int dim, int cin, int cout, int flags, void* stream);

/* Fused V2V tail: two 1x1x1 32->32 convs (+BN+ReLU) and the 1x1x1 32->cout3 output layer in one pass
 * (network/v2v.py:155-161 back_layers.1/.2 + output_layer :161,169).  in [B][D]^3[32]; out planar [B][cout3][D^3];
 * wpackN / bpackN come from se_conv3d_pack_f32 (ksize 1, cin_pad 32); cout3 <= 16.                           */
int se_pointwise_chain3_f32(const float* in, const float* wpack1, const float* bpack1,
                            const float* wpack2, const float* bpack2, const float* wpack3, const float* bpack3,
                            float* out, int batch, int dim, int cout3, void* stream);

/* se_pointwise_chain3_f32 with pass 1 of se_softargmax3d_f32 (mode 1: softmax) folded in: the logits are written to `out` as
 * before and, while still in registers, reduced to the per-chunk partial records in `scratch`
 * (se_softargmax3d_scratch_elems(batch * cout3) floats).  `coord` = [dim^3][3] voxel-centre coordinates.  Finish with
 * se_softargmax3d_finish_f32(out, scratch, ...).  Replaces the back_layers / output_layer chain of network/v2v.py:155-161
 * together with the first half of utils/op.py:83-96.  flags: 0, or SE_IN_QUAD: `in` is quad-planar [B][8][dim^3][4] (what
 * back_layers.0's last convolution writes with SE_OUT_QUAD). */
int se_pointwise_chain3_softargmax_f32(const float* in, const float* wpack1, const float* bpack1, const float* wpack2,
                                       const float* bpack2, const float* wpack3, const float* bpack3, float* out,
                                       const float* coord, float* scratch, int batch, int dim, int cout3, int flags, void* stream);

/* ConvTranspose3d(k=2, s=2) + folded BN + ReLU (+ skip).  Replaces Upsample3DBlock and the decoder
 * adds (network/v2v.py:55-67,124-137).  in [B][D]^3[cin] -> out [B][2D]^3[cout]; residual [B][2D]^3[cout] (channels-last).
 * flags: SE_EPI_RELU, SE_EPI_RES_PRE_RELU / SE_EPI_RES_POST_RELU, and SE_OUT_QUAD (cin -> cout = 64 -> 32 or 128 -> 64, D % 16 == 0
 * only, else SE_ERR_BAD_ARG): `out` is written quad-planar [B][cout/4][2D][2D][2D][4], the input layout of the 3x3x3 kernel behind it;
 * SE_RES_QUAD (with SE_OUT_QUAD and SE_EPI_RES_POST_RELU only): `residual` is quad-planar [B][cout/4][2D][2D][2D][4] too. */
int se_deconv3d_k2s2_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                         float* out, int batch, int dim, int cin, int cout, int flags, void* stream);

/* F.max_pool3d(kernel 2, stride 2) (network/v2v.py:46-52).  in [B][D]^3[C] -> out [B][D/2]^3[C]. */
int se_maxpool3d_2_f32(const float* in, float* out, int batch, int dim, int channels, void* stream);
/* Same with an octet-planar input [B][channels/8][dim^3][8] (the output of an SE_OUT_OCTET convolution); out is channels-last. */
int se_maxpool3d_2_octin_f32(const float* in, float* out, int batch, int dim, int channels, void* stream);

/* 3D soft-argmax.  Replaces op.integrate_tensor_3d_with_coordinates (utils/op.py:83-96).
 *   vol    [rows][voxels] float32 (rows = B*joints, planar logits, already multiplied by volume_multiplier)
 *   coord  [voxels][3] float32 voxel-centre coordinates
 *   out_vol[rows][voxels] softmax(vol) (mode 1) or relu(vol) (mode 0)
 *   joints [rows][3] = sum_n out_vol[n] * coord[n]
 *   scratch: se_softargmax3d_scratch_elems(rows) floats of workspace.                              */
int se_softargmax3d_f32(const float* vol, const float* coord, float* out_vol, float* joints,
                        float* scratch, int rows, int voxels, int mode, void* stream);

/* Pass 2 of se_softargmax3d_f32 alone (softmaxed volumes + joints from the partial records in `scratch`). */
int se_softargmax3d_finish_f32(const float* vol, const float* scratch, float* out_vol, float* joints, int rows, int voxels,
                               int mode, void* stream);
long long se_softargmax3d_scratch_elems(int rows);

/* Producers of the triplet-planar float32 V2V input [B][triplets_total][voxels][3] (SE_IN_PLANAR3; the 7^3 front layer
 * fetches its halo columns from ~5x fewer cache lines than from the channels-last record).  Same arithmetic, bit for bit, as
 * se_unproject_gather_f32 / se_voxelize_strided_f64.  The gather writes channels [0, channels) (channels = 16, 32 or 64) and
 * ZEROES the remaining slots of its last triplet; se_voxelize_planar3_f64 then only scatters 1.0 into slot `channel`
 * (it does not clear: call it after the gather, with channel in that cleared range, e.g. 32 for 32 feature channels).  */
int se_unproject_gather_planar3_f32(const float* feat, const int* idx, const float* w, float* out,
                                    int batch, int texels, int channels, int voxels, int triplets_total, void* stream);
int se_voxelize_planar3_f64(const float* depth, const double* ray_tab, float* buf, int batch, int depth_h, int depth_w,
                            int up, int pad_x, int volume_size, double cuboid_side,
                            int triplets_total, int channel, void* stream);

/* Producers of the fully PLANAR float32 V2V input [B][planes_total][voxels] (round 6: what se_conv3d_k7_fft_f32 reads).  Same
 * arithmetic, bit for bit, as se_unproject_gather_f32 / se_voxelize_strided_f64.  The gather writes planes [0, channels) (channels = 16,
 * 32 or 64) and ZEROES planes [channels, planes_total); se_voxelize_planar1_f64 then only scatters 1.0 into plane `channel` (call it
 * after the gather, with `channel` in that cleared range: 32 for 32 feature channels).  Reference call sites as for the planar3 pair. */
int se_unproject_gather_planar1_f32(const float* feat, const int* idx, const float* w, float* out,
                                    int batch, int texels, int channels, int voxels, int planes_total, void* stream);
int se_voxelize_planar1_f64(const float* depth, const double* ray_tab, float* buf, int batch, int depth_h, int depth_w,
                            int up, int pad_x, int volume_size, double cuboid_side,
                            int planes_total, int channel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The 7x7x7 front layer in the frequency domain (round 6; csrc/conv3d_fft7.hip).  Same reference call site as se_conv3d_f32 with
 * ksize 7: Basic3DBlock(33 | 32 -> 16, 7) = Conv3d(k 7, pad 3) + BatchNorm3d + ReLU, network/v2v.py:8-18 (built :147, run :166).
 * Three launches per chunk of samples: a 24^3 real-to-complex DFT of every (16^3-output tile, input channel), one complex GEMM over the
 * channels per frequency on the matrix cores, the inverse DFT of every (tile, output channel) with bias and ReLU.  float32 throughout;
 * results differ from the direct convolution by float32 rounding of the transforms (~1e-6 of max|y|).
 *   in   PLANAR float32 [B][cin][D][D][D]  (se_unproject_gather_planar1_f32 / se_voxelize_planar1_f64 write it)
 *   out  channels-last [B][D][D][D][16], or with SE_OUT_QUAD quad-planar [B][4][D][D][D][4]; flags: SE_EPI_RELU, SE_OUT_QUAD only
 *   hfrag  se_conv3d_k7_fft_packed_elems(cin, cout) floats from se_conv3d_k7_fft_pack_f32: the weight spectra (BatchNorm scale folded
 *          in: gamma / var / eps as for se_conv3d_pack_f32, NULL = no BatchNorm) in MFMA fragment order;  bpack: the folded bias that
 *          se_conv3d_pack_f32 writes (16 floats)
 *   workspace  the spectra of one chunk of samples: se_conv3d_k7_fft_workspace_elems(n, dim, cin) floats hold n samples (0.19 GB per
 *          sample at 64^3); the call walks the batch in chunks of as many samples as the workspace holds (>= 1, else SE_ERR_BAD_ARG).
 *          A workspace serves one stream at a time.
 * Shapes: cin = 33 (features + occupancy) or 32 (`with_scene: False`), cout = 16, dim % 16 == 0; the *_elems functions return -1 and the calls SE_ERR_BAD_ARG for anything else
 * (se_conv3d_f32 serves those). */
long long se_conv3d_k7_fft_packed_elems(int cin, int cout);
int se_conv3d_k7_fft_pack_f32(const float* w, const float* gamma, const float* var, float eps, float* hfrag, int cout, int cin,
                              void* stream);
long long se_conv3d_k7_fft_workspace_elems(int batch, int dim, int cin);
int se_conv3d_k7_fft_f32(const float* in, const float* hfrag, const float* bpack, float* out, int batch, int dim, int cin, int cout,
                         int flags, float* workspace, long long workspace_elems, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16-storage V2V (BASELINE config 3): activations and weights bfloat16 in HBM, float32 accumulation on
 * v_mfma_f32_16x16x32_bf16, float32 bias / BN shift, one round-to-nearest-even to bfloat16 per layer output.
 * Same reference call sites as the _f32 entry points above; `se_bf16` is the raw 16-bit pattern.
 * Volumes are channels-last [B][D][D][D][C] with C % 8 == 0 (a lane moves 8 channels = 16 bytes).
 * ------------------------------------------------------------------------------------------------ */
typedef unsigned short se_bf16;

/* Weight preparation (BN folded, MFMA A-fragment order [cout tile][k step][lane][8]).  A k step covers 4 groups of
 * (tap, 8 channels); the group order is [channel chunk of `chunk_octets` x 8 channels][tap][octet] with every chunk
 * padded to whole k steps.  chunk_octets: 2 for 3^3 convs (cin_pad % 16 == 0); 4 (cin_pad % 32 == 0) or 2 for 1^3 and
 * transposed convs; 1 for the 7^3 front layer (cin_pad = 40 / 72), whose taps are stored in the bank-conflict-free
 * pair order of the LDS kernel.
 * transposed != 0: ConvTranspose3d k2s2 weight [cin][cout][2][2][2], the 8 output parities take the place of taps.
 * cout % 32 == 0, or cout <= 16 (one tile).  wpack: se_conv3d_packed_elems_bf16(...) elements; bpack: float32,
 * cout rounded up to 16.                                                                                        */
int se_conv3d_pack_bf16(const float* w, const float* b, const float* gamma, const float* beta,
                        const float* mean, const float* var, float eps,
                        se_bf16* wpack, float* bpack,
                        int cout, int cin, int cin_pad, int ksize, int transposed, void* stream);
long long se_conv3d_packed_elems_bf16(int cout, int cin_pad, int ksize, int transposed);

/* EXPERIMENTAL (round 3; never on the float32 headline path - VoxelNetwork_depth.set_v2v_dtype("split_bf16") selects it): Conv3d
 * k = 3 + folded BN + epilogue on float32 channels-last tensors with SPLIT-bf16 arithmetic: every operand x = hi + lo
 * (hi = bf16(x), lo = bf16(x - hi)), a product = hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with float32 accumulation.
 * Replaces the same reference calls as se_conv3d_f32 with ksize 3 (network/v2v.py:21-43) for the shapes dim % 16 == 0,
 * cin_pad % 8 == 0, cout % 32 == 0.
 * se_conv3d_split3_pack: w = float32 [cout][cin][3][3][3] with the BatchNorm scale already folded in -> wsplit
 * (se_conv3d_split3_packed_elems bf16 elements; -1 for an unsupported shape): both halves in MFMA fragment order.
 * se_conv3d_k3_split3_f32: bpack = the folded float32 bias (as se_conv3d_pack_f32 writes it); flags: SE_EPI_RELU,
 * SE_EPI_RES_PRE_RELU / SE_EPI_RES_POST_RELU, SE_IN_OCTET / SE_OUT_OCTET / SE_RES_OCTET (octet-planar tensors, as for
 * se_conv3d_f32's 2-D Winograd shapes).  SE_ERR_BAD_ARG for anything else (nothing launched).                                 */
long long se_conv3d_split3_packed_elems(int cout, int cin_pad);
int se_conv3d_split3_pack(const float* w, se_bf16* wsplit, int cout, int cin, int cin_pad, void* stream);
int se_conv3d_k3_split3_f32(const float* in, const se_bf16* wsplit, const float* bpack, const float* residual, float* out,
                            int batch, int dim, int cin_pad, int cout, int flags, void* stream);

/* Conv3d k = 1, 3 or 7 + folded BN + epilogue (flags as se_conv3d_f32; SE_EPI_OUT_PLANAR is not supported: the
 * float32 planar logits come from se_pointwise_chain3_bf16).  in [B][D]^3[cin_pad] -> out [B][D]^3[cout].
 * k = 7 (the front layer) reads its input OCTET-PLANAR: in [B][cin_pad/8][D]^3[8] — the layer walks the input one
 * 8-channel octet at a time, and a channels-last record of 80 B would be fetched five times for 16 B each.      */
int se_conv3d_bf16(const se_bf16* in, const se_bf16* wpack, const float* bpack, const se_bf16* residual,
                   se_bf16* out, int batch, int dim, int cin_pad, int cout, int ksize, int flags, void* stream);

/* V2V tail as se_pointwise_chain3_f32: bfloat16 in [B][D]^3[32], float32 planar logits out [B][cout3][D^3]. */
int se_pointwise_chain3_bf16(const se_bf16* in, const se_bf16* wpack1, const float* bpack1,
                             const se_bf16* wpack2, const float* bpack2, const se_bf16* wpack3, const float* bpack3,
                             float* out, int batch, int dim, int cout3, void* stream);

/* se_pointwise_chain3_bf16 with pass 1 of se_softargmax3d_f32 (mode 1: softmax) folded in, as se_pointwise_chain3_softargmax_f32 does for
 * the float32 program (round 6): the float32 logits are written to `out` and, while in registers, reduced to the per-chunk partial records
 * in `scratch` (se_softargmax3d_scratch_elems(batch * cout3) floats); finish with se_softargmax3d_finish_f32.  network/v2v.py:155-161 +
 * the first half of utils/op.py:83-96. */
int se_pointwise_chain3_softargmax_bf16(const se_bf16* in, const se_bf16* wpack1, const float* bpack1, const se_bf16* wpack2,
                                        const float* bpack2, const se_bf16* wpack3, const float* bpack3, float* out,
                                        const float* coord, float* scratch, int batch, int dim, int cout3, void* stream);

int se_deconv3d_k2s2_bf16(const se_bf16* in, const se_bf16* wpack, const float* bpack, const se_bf16* residual,
                          se_bf16* out, int batch, int dim, int cin, int cout, int flags, void* stream);
int se_maxpool3d_2_bf16(const se_bf16* in, se_bf16* out, int batch, int dim, int channels, void* stream);

/* Producers of the bfloat16 V2V input, octet-planar buf [B][octs_total][voxels][8]: as se_unproject_gather_f32 /
 * se_voxelize_strided_f64.  The gather writes `channels` (% 8) channels starting at channel out_c_offset (% 8); the
 * voxeliser clears octet c_offset / 8 and writes the occupancy (1.0 = 0x3F80) into channel c_offset (% 8 == 0).
 * (with_intersection / scene_volumes inputs are assembled in float32 by the _f32 entry points and converted once.) */
int se_unproject_gather_bf16(const float* feat, const int* idx, const float* w, se_bf16* out,
                             int batch, int texels, int channels, int voxels, int octs_total, int out_c_offset,
                             void* stream);
int se_voxelize_strided_bf16(const float* depth, const double* ray_tab, se_bf16* buf, int batch, int depth_h,
                             int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                             int octs_total, int c_offset, void* stream);

/* Image pre-processing of the demo path on the device (dataset/demo_dataset.py:72-82, utils/data_transforms.py:38-72):
 * img BGR uint8 [B][height][width][3] -> crop crop_x columns each side -> exact 1/4 bilinear resize (= rounded mean of the
 * central 2x2 of every 4x4 block, cv2.resize INTER_LINEAR at scale 1/4) -> /255, -mean3[c], /std3[c] in float64 ->
 * out float32 [B][3][height/4][(width-2*crop_x)/4].  mean3 / std3: HOST pointers to 3 doubles (read at launch).   */
int se_preprocess_image_u8(const unsigned char* img, float* out, int batch, int height, int width, int crop_x,
                           const double* mean3, const double* std3, void* stream);

/* se_bias_act_nchw_f32 for a bfloat16 backbone (x, bias, residual, out bfloat16; float32 arithmetic; hw % 8 == 0). */
int se_bias_act_nchw_bf16(const se_bf16* x, const se_bf16* bias, const se_bf16* residual, se_bf16* out,
                          int batch, int channels, int hw, int relu, void* stream);

/* Which kernel family se_conv3d_f32 selects for a float32 convolution of this shape (bench.py prices the roofline with it):
 *   0 direct implicit GEMM (every product on the matrix cores),
 *   1 1-D Winograd F(4,3) along z (1/2 of the direct products), 2 2-D Winograd F(4,3) x F(2,3) along z, y (1/3),
 *   7 1-D Winograd along z for the 7x7x7 front layer: F(6,7) (12/42 of the direct products) when dim % 16 == 0, else F(4,7) (10/28).
 * Pure function of the arguments; no device access. */
int se_conv3d_f32_algo(int dim, int cin, int cout, int ksize);
/* Which kernel a launch of `batch` samples with these flags really runs on: se_conv3d_f32_algo()'s value, except
 *   3 = the F(4,3) x F(4,3) ping-pong kernel (a member of the 2-D Winograd family with the same fused forms; it executes 1/4 of the
 *       direct convolution's MFMAs, algo 2 executes 1/3).  Its planar layout is QUAD-planar (SE_IN_QUAD / SE_OUT_QUAD / SE_RES_QUAD):
 *       it takes a quad-planar input or a channels-last one with fewer than 32 channels; a channels-last input with >= 32 channels
 *       and every launch with an octet-planar flag stay on algo 2 (whose planar layout is octet-planar);
 *   0 for a 2-D Winograd shape with <= 4096 voxels in the batch when `flags` asks for none of the planar forms (such a call
 *       runs on the in-workgroup split-K kernel).
 * `flags`: the SE_IN_* / SE_OUT_* / SE_RES_* layout bits of the launch (others ignored).  A caller that wants planar hand-overs asks
 * with the quad bits first: 3 = use them; otherwise the octet bits (2 = use those).  bench.py prices every launch with it. */
int se_conv3d_f32_variant(int batch, int dim, int cin, int cout, int ksize, int flags);

#ifdef SE_DEVTOOLS
/* Development builds only (csrc/build.sh --devtools; absent from the production library): A/B kernel selection for
 * tools/bench_conv.py and the cycle-stamp diagnostics.  The selector is thread-local. */
void se_debug_set_variant(int variant);
/* u64 device buffer [workgroups][8 waves][6]; non-NULL switches the Winograd conv to its cycle-stamp build. */
void se_debug_set_stamp_buffer(void* device_buffer);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SCENEEGO_HIP_H */
