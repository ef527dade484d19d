; opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes="print<cost-model>" -cost-kind=latency -disable-output cost_model_probe.ll
; which instructions the AMDGPU latency cost model prices at zero (StructurizeCFG hoists only those; tools/structurize_lint.py)
target triple = "amdgcn-amd-amdhsa"
%S = type { float, float, float }
define amdgpu_kernel void @k(ptr addrspace(1) %p, float %a, float %b, i32 %i, <2 x float> %v, <4 x float> %w, i64 %l, %S %s, half %h, ptr %g, <2 x i32> %vi, double %d, i16 %sh) {
  %bc1 = bitcast float %a to i32
  %bc2 = bitcast i32 %i to float
  %bc3 = bitcast <2 x float> %v to <2 x i32>
  %bc4 = bitcast <2 x float> %v to i64
  %bc5 = bitcast <2 x float> %v to double
  %ie1 = insertelement <2 x float> <float 1.0, float poison>, float %a, i64 1
  %ie2 = insertelement <2 x float> %v, float %a, i32 0
  %ie3 = insertelement <4 x float> %w, float %a, i32 %i
  %ie4 = insertelement <2 x half> poison, half %h, i32 1
  %ee1 = extractelement <2 x float> %v, i64 0
  %ee2 = extractelement <4 x float> %w, i32 %i
  %ee3 = extractelement <2 x i32> %vi, i32 1
  %iv1 = insertvalue %S %s, float %a, 1
  %ev1 = extractvalue %S %s, 2
  %sv1 = shufflevector <2 x float> %v, <2 x float> poison, <2 x i32> zeroinitializer
  %sv2 = shufflevector <2 x float> %v, <2 x float> poison, <2 x i32> <i32 1, i32 0>
  %sv3 = shufflevector <4 x float> %w, <4 x float> poison, <2 x i32> <i32 0, i32 1>
  %sv4 = shufflevector <4 x float> %w, <4 x float> poison, <2 x i32> <i32 1, i32 3>
  %sv5 = shufflevector <2 x float> %v, <2 x float> %v, <2 x i32> <i32 1, i32 3>
  %fn1 = fneg float %a
  %fn2 = fneg <2 x float> %v
  %fa1 = call float @llvm.fabs.f32(float %a)
  %fa2 = call <2 x float> @llvm.fabs.v2f32(<2 x float> %v)
  %fr1 = freeze float %a
  %tr1 = trunc i64 %l to i32
  %tr2 = trunc i32 %i to i16
  %ze1 = zext i32 %i to i64
  %ze2 = zext i16 %sh to i32
  %se1 = sext i32 %i to i64
  %fe1 = fpext float %a to double
  %fe2 = fpext half %h to float
  %ft1 = fptrunc double %d to float
  %gp1 = getelementptr inbounds i8, ptr addrspace(1) %p, i64 16
  %gp2 = getelementptr inbounds float, ptr addrspace(1) %p, i32 %i
  %gp3 = getelementptr inbounds float, ptr %g, i64 %l
  %pi1 = ptrtoint ptr addrspace(1) %p to i64
  %ip1 = inttoptr i64 %l to ptr addrspace(1)
  %ac1 = addrspacecast ptr addrspace(1) %p to ptr
  %cn1 = call float @llvm.canonicalize.f32(float %a)
  %rf1 = call i32 @llvm.amdgcn.readfirstlane.i32(i32 %i)
  %sel = select i1 true, float %a, float %b
  %add = add i32 %i, 1
  %fm = fmul float %a, %b
  %an = and i32 %i, 255
  %sh1 = shl i32 %i, 2
  %cmp = fcmp olt float %a, %b
  %cs = call float @llvm.copysign.f32(float %a, float %b)
  %mx = call float @llvm.maxnum.f32(float %a, float %b)
  %u2f = uitofp i32 %i to float
  %f2i = fptosi float %a to i32
  ret void
}
declare float @llvm.fabs.f32(float)
declare <2 x float> @llvm.fabs.v2f32(<2 x float>)
declare float @llvm.canonicalize.f32(float)
declare i32 @llvm.amdgcn.readfirstlane.i32(i32)
declare float @llvm.copysign.f32(float, float)
declare float @llvm.maxnum.f32(float, float)
