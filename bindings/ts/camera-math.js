'use strict';
/*
 * camera-math.js -- the arithmetic behind the 272-byte camera block (src/camera/camera.ts:23-56, 138-205) shared by trainer.js, loaders.js,
 * viewer.js and synth.js: wgpu-matrix's mat4.inverse (the reference's dependency, version 3.2.0, not vendored in the reference tree: its
 * published cofactor algorithm restated), get_projection_matrix, and the block of a view re-targeted to another canvas size.
 */

/** wgpu-matrix 3.2.0 mat4.inverse (the reference's dependency; cofactor expansion in binary64 on the Float32Array's values). */
function mat4Inverse(m) {
  const m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  const m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
  const tmp0 = m22 * m33, tmp1 = m32 * m23, tmp2 = m12 * m33, tmp3 = m32 * m13, tmp4 = m12 * m23, tmp5 = m22 * m13;
  const tmp6 = m02 * m33, tmp7 = m32 * m03, tmp8 = m02 * m23, tmp9 = m22 * m03, tmp10 = m02 * m13, tmp11 = m12 * m03;
  const tmp12 = m20 * m31, tmp13 = m30 * m21, tmp14 = m10 * m31, tmp15 = m30 * m11, tmp16 = m10 * m21, tmp17 = m20 * m11;
  const tmp18 = m00 * m31, tmp19 = m30 * m01, tmp20 = m00 * m21, tmp21 = m20 * m01, tmp22 = m00 * m11, tmp23 = m10 * m01;
  const t0 = (tmp0 * m11 + tmp3 * m21 + tmp4 * m31) - (tmp1 * m11 + tmp2 * m21 + tmp5 * m31);
  const t1 = (tmp1 * m01 + tmp6 * m21 + tmp9 * m31) - (tmp0 * m01 + tmp7 * m21 + tmp8 * m31);
  const t2 = (tmp2 * m01 + tmp7 * m11 + tmp10 * m31) - (tmp3 * m01 + tmp6 * m11 + tmp11 * m31);
  const t3 = (tmp5 * m01 + tmp8 * m11 + tmp11 * m21) - (tmp4 * m01 + tmp9 * m11 + tmp10 * m21);
  const d = 1 / (m00 * t0 + m10 * t1 + m20 * t2 + m30 * t3);
  const o = new Float32Array(16);
  o[0] = d * t0; o[1] = d * t1; o[2] = d * t2; o[3] = d * t3;
  o[4] = d * ((tmp1 * m10 + tmp2 * m20 + tmp5 * m30) - (tmp0 * m10 + tmp3 * m20 + tmp4 * m30));
  o[5] = d * ((tmp0 * m00 + tmp7 * m20 + tmp8 * m30) - (tmp1 * m00 + tmp6 * m20 + tmp9 * m30));
  o[6] = d * ((tmp3 * m00 + tmp6 * m10 + tmp11 * m30) - (tmp2 * m00 + tmp7 * m10 + tmp10 * m30));
  o[7] = d * ((tmp4 * m00 + tmp9 * m10 + tmp10 * m20) - (tmp5 * m00 + tmp8 * m10 + tmp11 * m20));
  o[8] = d * ((tmp12 * m13 + tmp15 * m23 + tmp16 * m33) - (tmp13 * m13 + tmp14 * m23 + tmp17 * m33));
  o[9] = d * ((tmp13 * m03 + tmp18 * m23 + tmp21 * m33) - (tmp12 * m03 + tmp19 * m23 + tmp20 * m33));
  o[10] = d * ((tmp14 * m03 + tmp19 * m13 + tmp22 * m33) - (tmp15 * m03 + tmp18 * m13 + tmp23 * m33));
  o[11] = d * ((tmp17 * m03 + tmp20 * m13 + tmp23 * m23) - (tmp16 * m03 + tmp21 * m13 + tmp22 * m23));
  o[12] = d * ((tmp14 * m22 + tmp17 * m32 + tmp13 * m12) - (tmp16 * m32 + tmp12 * m12 + tmp15 * m22));
  o[13] = d * ((tmp20 * m32 + tmp12 * m02 + tmp19 * m22) - (tmp18 * m22 + tmp21 * m32 + tmp13 * m02));
  o[14] = d * ((tmp18 * m12 + tmp23 * m32 + tmp15 * m02) - (tmp22 * m32 + tmp14 * m02 + tmp19 * m12));
  o[15] = d * ((tmp22 * m22 + tmp16 * m02 + tmp21 * m12) - (tmp20 * m12 + tmp23 * m22 + tmp17 * m02));
  return o;
}

/** get_projection_matrix (src/camera/camera.ts:29-56), column-major after its transpose. */
function projectionMatrix(znear, zfar, fovX, fovY) {
  const tanY = Math.tan(fovY / 2), tanX = Math.tan(fovX / 2);
  const top = tanY * znear, right = tanX * znear;
  const p = new Float32Array(16);
  p[0] = 2 * znear / (2 * right);
  p[5] = -2 * znear / (2 * top);
  p[10] = zfar / (zfar - znear);
  p[11] = 1;
  p[14] = -(zfar * znear) / (zfar - znear);
  return p;
}

/** Camera.set_preset + on_update_canvas + update_buffer (camera.ts:138-205) for a view given as its 68-float block, on a canvas of
 *  width x height: the pose is kept, fovY = 2 atan(height_view / (2 fy_view)), focal = 0.5 height / tan(fovY / 2). */
function cameraBlockFor(block, width, height) {
  const fovY = 2 * Math.atan(block[65] / (2 * block[67]));
  const focal = 0.5 * height / Math.tan(fovY * 0.5);
  const fovX = 2 * Math.atan(width / (2 * focal));
  const out = new Float32Array(68);
  out.set(block.subarray(0, 16), 0);
  out.set(projectionMatrix(0.01, 100, fovX, fovY), 32);
  out.set(mat4Inverse(out.subarray(0, 16)), 16);
  out.set(mat4Inverse(out.subarray(32, 48)), 48);
  out[64] = width; out[65] = height; out[66] = focal; out[67] = focal;
  return out;
}

module.exports = { mat4Inverse, projectionMatrix, cameraBlockFor };
