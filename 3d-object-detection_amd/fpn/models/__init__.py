"""Model builders: counterparts of models/frustum_pointnets_v1.py,
models/FPointNet_Keras.py, models/FPointNet_Bayes.py and models/frustum_pointnet.py."""
