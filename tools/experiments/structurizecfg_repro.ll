; StructurizeCFG defect of hipcc 7.2 / AMD clang 22 reduced to one loop (profiles/r03_slp_root_cause.md).
;   /opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes=structurizecfg -S structurizecfg_repro.ll
; Input: a lane that takes entry -> roulette -> survive -> cont -> exit leaves with %ray = %pair.
; Output: `Flow: %0 = phi <2 x float> [ %old, %Flow1 ], [ %pair, %entry ]` and `exit: %ray = phi ... [ %0, %Flow2 ]`: that lane now
; leaves with %old.  Make %pair cost something (e.g. `fadd <2 x float> %pw, %pw` after the insertelement) and the output is right
; (`Flow2: phi [ %pair, %cont ], [ %old, %Flow ]`): the pass hoists only instructions its cost model prices at zero.
target triple = "amdgcn-amd-amdhsa"
define amdgpu_kernel void @rr(ptr addrspace(1) %p, ptr addrspace(1) %in, float %t0, float %q, i32 %n) {
pre:
  %tid = call i32 @llvm.amdgcn.workitem.id.x()
  %tf = uitofp i32 %tid to float
  br label %head

head:
  %i = phi i32 [ 0, %pre ], [ %i1, %exit ]
  %old = phi <2 x float> [ zeroinitializer, %pre ], [ %ray, %exit ]
  %t = phi float [ %t0, %pre ], [ %tn, %exit ]
  %g = getelementptr inbounds float, ptr addrspace(1) %in, i32 %i
  %wi = load float, ptr addrspace(1) %g, align 4
  %alive = fcmp ogt float %wi, %tf
  br i1 %alive, label %entry, label %exit

entry:
  %td = fadd float %t, %tf
  %c = fcmp olt float %td, 1.000000e+00
  br i1 %c, label %cont, label %roulette

roulette:
  %qd = fadd float %q, %wi
  %s = fcmp olt float %qd, 5.000000e-01
  br i1 %s, label %survive, label %exit

survive:
  %t2 = fdiv float %t, %q
  br label %cont

cont:
  %tt = phi float [ %t, %entry ], [ %t2, %survive ]
  %u = fmul float %tt, %wi
  %pair = insertelement <2 x float> <float 0x47EFFFFFE0000000, float poison>, float %wi, i64 1
  br label %exit

exit:
  %ray = phi <2 x float> [ %pair, %cont ], [ %old, %roulette ], [ %old, %head ]
  %tn = phi float [ %u, %cont ], [ %t, %roulette ], [ %t, %head ]
  %i1 = add i32 %i, 1
  %more = icmp ult i32 %i1, %n
  br i1 %more, label %head, label %done

done:
  store <2 x float> %ray, ptr addrspace(1) %p, align 8
  ret void
}
declare i32 @llvm.amdgcn.workitem.id.x()
