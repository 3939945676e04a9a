5417:                                             ; preds = %5409
  %5418 = trunc i64 %5411 to i32
  %5419 = icmp eq i32 %5418, 3
  %5420 = select i1 %5419, i32 %5184, i32 %5410
  %5421 = inttoptr i32 %5420 to ptr addrspace(3)
  %5422 = load <4 x float>, ptr addrspace(3) %5421, align 16, !tbaa !49
  %5423 = add i32 %5420, 16
  %5424 = inttoptr i32 %5423 to ptr addrspace(3)
  %5425 = load <4 x float>, ptr addrspace(3) %5424, align 16, !tbaa !49
  %5426 = add i32 %5420, 32
  %5427 = inttoptr i32 %5426 to ptr addrspace(3)
  %5428 = load <4 x float>, ptr addrspace(3) %5427, align 16, !tbaa !49
  %5429 = add i32 %5420, 48
  %5430 = inttoptr i32 %5429 to ptr addrspace(3)
  %5431 = load <4 x float>, ptr addrspace(3) %5430, align 16, !tbaa !49
  %5432 = shufflevector <4 x float> %5416, <4 x float> poison, <2 x i32> <i32 0, i32 1>
  %5433 = shufflevector <4 x float> %5415, <4 x float> poison, <2 x i32> <i32 0, i32 1>
  %5434 = shufflevector <4 x float> %5414, <4 x float> poison, <2 x i32> <i32 0, i32 1>
  %5435 = shufflevector <4 x float> %5413, <4 x float> poison, <2 x i32> <i32 0, i32 1>
  %5436 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5435, <2 x float> %5162, <2 x float> %5434)
  %5437 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5436, <2 x float> %5162, <2 x float> %5433)
  %5438 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5437, <2 x float> %5162, <2 x float> %5432)
  %5439 = shufflevector <4 x float> %5416, <4 x float> poison, <2 x i32> <i32 2, i32 3>
  %5440 = shufflevector <4 x float> %5415, <4 x float> poison, <2 x i32> <i32 2, i32 3>
  %5441 = shufflevector <4 x float> %5414, <4 x float> poison, <2 x i32> <i32 2, i32 3>
  %5442 = shufflevector <4 x float> %5413, <4 x float> poison, <2 x i32> <i32 2, i32 3>
  %5443 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5442, <2 x float> %5162, <2 x float> %5441)
  %5444 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5443, <2 x float> %5162, <2 x float> %5440)
  %5445 = call <2 x float> @llvm.fma.v2f32(<2 x float> %5444, <2 x float> %5162, <2 x float> %5439)
  %5446 = extractelement <2 x float> %5438, i64 1
  %5447 = icmp slt i32 %4595, 1
  br i1 %5447, label %5450, label %5448

5450:                                             ; preds = %5417
  %5451 = icmp eq i32 %4595, 0
  br i1 %5451, label %5452, label %5499

5452:                                             ; preds = %5450
  %5453 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5454 = fsub <2 x float> %5445, %5453
  %5455 = extractelement <2 x float> %5454, i64 0
  %5456 = fsub <2 x float> %5453, %5438
  %5457 = extractelement <2 x float> %5456, i64 0
  %5458 = fdiv float %5457, %4597
  %5459 = fmul float %4600, %5458
  %5460 = call noundef float @llvm.fma.f32(float %4599, float %5455, float %5459)
  %5461 = shufflevector <2 x float> %5445, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5462 = fsub <2 x float> %5461, %5445
  %5463 = extractelement <2 x float> %5462, i64 0
  %5464 = fdiv float %5463, %4601
  %5465 = fmul <2 x float> %5454, %5130
  %5466 = extractelement <2 x float> %5465, i64 0
  %5467 = call noundef float @llvm.fma.f32(float %4602, float %5464, float %5466)
  %5468 = insertelement <2 x float> poison, float %5460, i64 0
  %5469 = shufflevector <2 x float> %5468, <2 x float> %5454, <2 x i32> <i32 0, i32 2>
  %5470 = insertelement <2 x float> %5454, float %5467, i64 1
  %5471 = fsub <2 x float> %5469, %5470
  %5472 = fadd <2 x float> %5454, %5471
  %5473 = extractelement <2 x float> %5472, i64 0
  %5474 = fadd <2 x float> %5471, %5471
  %5475 = shufflevector <2 x float> %5471, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5476 = fsub <2 x float> %5475, %5474
  %5477 = extractelement <2 x float> %5476, i64 0
  %5478 = fsub <2 x float> %5471, %5475
  %5479 = extractelement <2 x float> %5478, i64 0
  %5480 = call noundef float @llvm.fma.f32(float %5479, float %4598, float %5477)
  %5481 = call noundef float @llvm.fma.f32(float %5480, float %4598, float %5473)
  %5482 = call noundef float @llvm.fma.f32(float %5481, float %4598, float %5446)
  br label %5546

5499:                                             ; preds = %5448, %5450
  %5500 = extractelement <2 x float> %5445, i64 1
  %5501 = extractelement <2 x float> %5445, i64 0
  %5502 = shufflevector <2 x float> %5445, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5503 = fsub <2 x float> %5502, %5445
  %5504 = extractelement <2 x float> %5503, i64 0
  %5505 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5506 = fsub <2 x float> %5445, %5505
  %5507 = extractelement <2 x float> %5506, i64 0
  %5508 = fdiv float %5507, %4597
  %5509 = fmul float %4600, %5508
  %5510 = call noundef float @llvm.fma.f32(float %4599, float %5504, float %5509)
  %5511 = insertelement <2 x float> poison, float %5510, i64 0
  %5512 = shufflevector <2 x float> %5511, <2 x float> %5503, <2 x i32> <i32 0, i32 2>
  %5513 = extractelement <2 x float> %5512, i64 1
  br label %5514

5514:                                             ; preds = %5499, %5483
  %5515 = phi <2 x float> [ %5511, %5499 ], [ %5498, %5483 ]
  %5516 = phi float [ %5446, %5483 ], [ %5501, %5499 ]
  %5517 = phi float [ %5484, %5483 ], [ %5500, %5499 ]
  %5518 = phi float [ %5495, %5483 ], [ %5510, %5499 ]
  %5519 = phi float [ %5497, %5483 ], [ %5513, %5499 ]
  %5520 = fmul float %5519, 2.000000e+00
  %5521 = fsub float %5520, %5518
  br i1 %5134, label %5528, label %5522

5528:                                             ; preds = %5514
  %5529 = insertelement <2 x float> %5515, float %5519, i64 1
  %5530 = shufflevector <2 x float> %5529, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5531 = insertelement <2 x float> %5530, float %5521, i64 1
  %5532 = fsub <2 x float> %5529, %5531
  %5533 = fadd <2 x float> %5530, %5532
  %5534 = extractelement <2 x float> %5533, i64 0
  %5535 = fadd <2 x float> %5532, %5532
  %5536 = shufflevector <2 x float> %5532, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5537 = fsub <2 x float> %5536, %5535
  %5538 = extractelement <2 x float> %5537, i64 0
  %5539 = fsub <2 x float> %5532, %5536
  %5540 = extractelement <2 x float> %5539, i64 0
  %5541 = call noundef float @llvm.fma.f32(float %5540, float %4598, float %5538)
  %5542 = call noundef float @llvm.fma.f32(float %5541, float %4598, float %5534)
  %5543 = call noundef float @llvm.fma.f32(float %5542, float %4598, float %5516)
  br label %5522

5522:                                             ; preds = %5528, %5514
  %5523 = phi float [ %5543, %5528 ], [ poison, %5514 ]
  %5524 = phi i1 [ false, %5528 ], [ true, %5514 ]
  br i1 %5524, label %5525, label %5544

5525:                                             ; preds = %5522
  %5526 = fmul float %5521, %5163
  %5527 = fadd float %5517, %5526
  br label %5544

5544:                                             ; preds = %5525, %5522
  %5545 = phi float [ %5527, %5525 ], [ %5523, %5522 ]
  br label %5546

5448:                                             ; preds = %5417
  %5449 = icmp eq i32 %4595, 1
  br i1 %5449, label %5483, label %5499

5483:                                             ; preds = %5448
  %5484 = extractelement <2 x float> %5438, i64 0
  %5485 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 poison, i32 0>
  %5486 = fsub <2 x float> %5485, %5438
  %5487 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5488 = fsub <2 x float> %5445, %5487
  %5489 = extractelement <2 x float> %5488, i64 0
  %5490 = fdiv float %5489, %4597
  %5491 = fsub <2 x float> %5487, %5438
  %5492 = fmul <2 x float> %5165, %5491
  %5493 = extractelement <2 x float> %5492, i64 0
  %5494 = call noundef float @llvm.fma.f32(float %4599, float %5490, float %5493)
  %5495 = fneg float %5494
  %5496 = insertelement <2 x float> %5486, float %5495, i64 0
  %5497 = extractelement <2 x float> %5496, i64 1
  %5498 = insertelement <2 x float> poison, float %5495, i64 0
  br label %5514
