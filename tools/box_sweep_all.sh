python tools/box_sweep.py 260x346_1_512 4,4,32 8,4,16 4,8,16 8,8,8 16,4,8 2,8,32
python tools/box_sweep.py 130x173_1_512 8,4,16 4,8,16 4,4,32 8,8,8
python tools/box_sweep.py 65x87_1_256 16,2,8 8,4,8 4,4,16 8,2,16
python tools/box_sweep.py 33x44_1_256 1,11,23 2,11,11 1,11,22
python tools/box_sweep.py 130x173_2_128 16,1,8 8,1,16 4,2,16 8,2,8
python tools/box_sweep.py 65x87_2_128 16,1,8 8,1,16 4,2,16 8,2,8
