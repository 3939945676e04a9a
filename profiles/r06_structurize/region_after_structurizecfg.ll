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
  %5447 = icmp sge i32 %4595, 1
  %5448 = extractelement <2 x float> %5445, i64 0
  %5449 = extractelement <2 x float> %5445, i64 1
  br i1 %5447, label %5450, label %5494

5450:                                             ; preds = %5417
  %5451 = icmp eq i32 %4595, 1
  br i1 %5451, label %5502, label %5452

5502:                                             ; preds = %5450
  %5503 = extractelement <2 x float> %5438, i64 0
  %5504 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 poison, i32 0>
  %5505 = fsub <2 x float> %5504, %5438
  %5506 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5507 = fsub <2 x float> %5445, %5506
  %5508 = extractelement <2 x float> %5507, i64 0
  %5509 = fdiv float %5508, %4597
  %5510 = fsub <2 x float> %5506, %5438
  %5511 = fmul <2 x float> %5165, %5510
  %5512 = extractelement <2 x float> %5511, i64 0
  %5513 = call noundef float @llvm.fma.f32(float %4599, float %5509, float %5512)
  %5514 = fneg float %5513
  %5515 = insertelement <2 x float> %5505, float %5514, i64 0
  %5516 = extractelement <2 x float> %5515, i64 1
  %5517 = insertelement <2 x float> poison, float %5514, i64 0
  br label %5452

5452:                                             ; preds = %5502, %5450
  %5453 = phi float [ %5516, %5502 ], [ poison, %5450 ]
  %5454 = phi float [ %5514, %5502 ], [ poison, %5450 ]
  %5455 = phi float [ %5503, %5502 ], [ poison, %5450 ]
  %5456 = phi <2 x float> [ %5517, %5502 ], [ poison, %5450 ]
  %5457 = phi i1 [ true, %5502 ], [ false, %5450 ]
  %5458 = phi i1 [ false, %5502 ], [ true, %5450 ]
  br label %5494

5494:                                             ; preds = %5452, %5417
  %5495 = phi float [ %5453, %5452 ], [ poison, %5417 ]
  %5496 = phi float [ %5454, %5452 ], [ poison, %5417 ]
  %5497 = phi float [ %5455, %5452 ], [ poison, %5417 ]
  %5498 = phi <2 x float> [ %5456, %5452 ], [ poison, %5417 ]
  %5499 = phi i1 [ %5457, %5452 ], [ false, %5417 ]
  %5500 = phi i1 [ %5458, %5452 ], [ false, %5417 ]
  %5501 = phi i1 [ false, %5452 ], [ true, %5417 ]
  br i1 %5501, label %5459, label %5518

5459:                                             ; preds = %5494
  %5460 = icmp ne i32 %4595, 0
  br label %5518

5518:                                             ; preds = %5459, %5494
  %5519 = phi float [ %5449, %5459 ], [ %5497, %5494 ]
  %5520 = phi float [ %5448, %5459 ], [ %5446, %5494 ]
  %5521 = phi i1 [ true, %5459 ], [ false, %5494 ]
  %5522 = phi i1 [ %5460, %5459 ], [ %5500, %5494 ]
  br i1 %5522, label %5523, label %5536

5523:                                             ; preds = %5518
  %5524 = shufflevector <2 x float> %5445, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5525 = fsub <2 x float> %5524, %5445
  %5526 = extractelement <2 x float> %5525, i64 0
  %5527 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5528 = fsub <2 x float> %5445, %5527
  %5529 = extractelement <2 x float> %5528, i64 0
  %5530 = fdiv float %5529, %4597
  %5531 = fmul float %4600, %5530
  %5532 = call noundef float @llvm.fma.f32(float %4599, float %5526, float %5531)
  %5533 = insertelement <2 x float> poison, float %5532, i64 0
  %5534 = shufflevector <2 x float> %5533, <2 x float> %5525, <2 x i32> <i32 0, i32 2>
  %5535 = extractelement <2 x float> %5534, i64 1
  br label %5536

5536:                                             ; preds = %5523, %5518
  %5537 = phi float [ %5535, %5523 ], [ %5495, %5518 ]
  %5538 = phi float [ %5532, %5523 ], [ %5496, %5518 ]
  %5539 = phi <2 x float> [ %5533, %5523 ], [ %5498, %5518 ]
  %5540 = phi i1 [ false, %5523 ], [ %5521, %5518 ]
  %5541 = phi i1 [ true, %5523 ], [ %5499, %5518 ]
  br i1 %5541, label %5542, label %5461

5542:                                             ; preds = %5536
  %5543 = fmul float %5537, 2.000000e+00
  %5544 = fsub float %5543, %5538
  br i1 %5134, label %5551, label %5545

5551:                                             ; preds = %5542
  %5552 = insertelement <2 x float> %5539, float %5537, i64 1
  %5553 = shufflevector <2 x float> %5552, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5554 = insertelement <2 x float> %5553, float %5544, i64 1
  %5555 = fsub <2 x float> %5552, %5554
  %5556 = fadd <2 x float> %5553, %5555
  %5557 = extractelement <2 x float> %5556, i64 0
  %5558 = fadd <2 x float> %5555, %5555
  %5559 = shufflevector <2 x float> %5555, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5560 = fsub <2 x float> %5559, %5558
  %5561 = extractelement <2 x float> %5560, i64 0
  %5562 = fsub <2 x float> %5555, %5559
  %5563 = extractelement <2 x float> %5562, i64 0
  %5564 = call noundef float @llvm.fma.f32(float %5563, float %4598, float %5561)
  %5565 = call noundef float @llvm.fma.f32(float %5564, float %4598, float %5557)
  %5566 = call noundef float @llvm.fma.f32(float %5565, float %4598, float %5520)
  br label %5545

5545:                                             ; preds = %5551, %5542
  %5546 = phi float [ %5566, %5551 ], [ poison, %5542 ]
  %5547 = phi i1 [ false, %5551 ], [ true, %5542 ]
  br i1 %5547, label %5548, label %5567

5548:                                             ; preds = %5545
  %5549 = fmul float %5544, %5163
  %5550 = fadd float %5519, %5549
  br label %5567

5567:                                             ; preds = %5548, %5545
  %5568 = phi float [ %5550, %5548 ], [ %5546, %5545 ]
  br label %5461

5461:                                             ; preds = %5567, %5536
  %5462 = phi float [ %5568, %5567 ], [ poison, %5536 ]
  br i1 %5540, label %5463, label %5569

5463:                                             ; preds = %5461
  %5464 = shufflevector <2 x float> %5438, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5465 = fsub <2 x float> %5445, %5464
  %5466 = extractelement <2 x float> %5465, i64 0
  %5467 = fsub <2 x float> %5464, %5438
  %5468 = extractelement <2 x float> %5467, i64 0
  %5469 = fdiv float %5468, %4597
  %5470 = fmul float %4600, %5469
  %5471 = call noundef float @llvm.fma.f32(float %4599, float %5466, float %5470)
  %5472 = shufflevector <2 x float> %5445, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5473 = fsub <2 x float> %5472, %5445
  %5474 = extractelement <2 x float> %5473, i64 0
  %5475 = fdiv float %5474, %4601
  %5476 = fmul <2 x float> %5465, %5130
  %5477 = extractelement <2 x float> %5476, i64 0
  %5478 = call noundef float @llvm.fma.f32(float %4602, float %5475, float %5477)
  %5479 = insertelement <2 x float> poison, float %5471, i64 0
  %5480 = shufflevector <2 x float> %5479, <2 x float> %5465, <2 x i32> <i32 0, i32 2>
  %5481 = insertelement <2 x float> %5465, float %5478, i64 1
  %5482 = fsub <2 x float> %5480, %5481
  %5483 = fadd <2 x float> %5465, %5482
  %5484 = extractelement <2 x float> %5483, i64 0
  %5485 = fadd <2 x float> %5482, %5482
  %5486 = shufflevector <2 x float> %5482, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %5487 = fsub <2 x float> %5486, %5485
  %5488 = extractelement <2 x float> %5487, i64 0
  %5489 = fsub <2 x float> %5482, %5486
  %5490 = extractelement <2 x float> %5489, i64 0
  %5491 = call noundef float @llvm.fma.f32(float %5490, float %4598, float %5488)
  %5492 = call noundef float @llvm.fma.f32(float %5491, float %4598, float %5484)
  %5493 = call noundef float @llvm.fma.f32(float %5492, float %4598, float %5446)
  br label %5569
